# tiled whole-block cross-attention kernel (AF_XATTN_TILED) against the wave-owns-tokens form and the three-launch path: parity tests, then tools/bench_xattn.py
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider -k "xattn_fused" 2>&1 | tail -3
for t in 0 1 0 1; do AF_XATTN_TILED=$t python $R/tools/bench_xattn.py 2>&1 | grep -v amdgpu | tail -4; done
