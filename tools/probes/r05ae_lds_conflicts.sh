# r05ae: LDS bank-conflict share per kernel over a denoise step and a Stage-1 training leg (which other kernel has a swizzle that assumes consecutive-lane groups?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05ae_lds_conflicts.txt
: > $OUT
rm -rf /tmp/pd /tmp/pt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d /tmp/pd -- python3 $R/bench.py --mode denoise --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pd.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d /tmp/pt -- python3 $R/bench.py --mode train --train-steps 1 --train-warmup 1 --no-cpu-baseline --no-roofline > /tmp/pt.log 2>&1
for leg in pd pt; do
  echo "##### leg $leg" >> $OUT
  for pat in af_gemm3w_kernel af_gemm3_kernel af_gemm_kernel af_conv3h af_ff320 af_xattn320t af_gn_proj320 af_attn2_kernel af_attn_kernel af_xattn_kernel attn_bwd_dq attn_bwd_dkv xattn_ gn_apply gn_partial gn_small gn_pair gn_bwd layernorm transpose af_splitk; do
    echo "=== $pat" >> $OUT
    python3 $R/tools/pmc_kernel.py "$pat" $(find /tmp/$leg -name "*_results.db") 2>&1 | grep "SQ_LDS_BANK_CONFLICT\|SQ_LDS_IDX_ACTIVE" >> $OUT
  done
done
python3 - <<'PY' >> $OUT
import re
txt=open('/root/repo/gpurun_out/r05ae_lds_conflicts.txt').read()
print("##### summary: conflict cycles / LDS-active cycles per kernel pattern")
leg=None; pat=None; v={}
for l in txt.splitlines():
    if l.startswith('##### leg'): leg=l.split()[-1]
    elif l.startswith('=== '): pat=l[4:]
    else:
        m=re.match(r'(\w+)\s+n=\s*(\d+) avg=\s*([\d.]+)',l)
        if m: v[(leg,pat,m.group(1))]=(int(m.group(2)),float(m.group(3)))
for (leg,pat,c),(n,a) in sorted(v.items()):
    if c=='SQ_LDS_BANK_CONFLICT':
        act=v.get((leg,pat,'SQ_LDS_IDX_ACTIVE'),(0,0))[1]
        if act>0: print(f"{leg} {pat:20s} launches {n:5d}  conflicts/active {a/act:.3f}   active per launch {act:.3e}")
PY
tail -45 $OUT
