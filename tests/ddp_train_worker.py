"""Worker of tests/test_hip_train.py::test_data_parallel_step_two_ranks_equals_accumulated_single_process (launched with
torch.distributed.run, 2 ranks on one GPU, gloo) and ::test_rccl_world1_train_step_under_launcher (AF_DDP_BACKEND=nccl, ONE
rank: RCCL refuses two ranks on one device).  Rank r trains on micro-batch r with accumulate_grad_batches=1."""
import os
import sys

import torch
import torch.distributed as dist


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("AF_DDP_BACKEND", "gloo")
    dev = torch.device("cuda:0")
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaface_dev_amd import rng
    from adaface_dev_amd.distributed import GradReducer
    from trainer_util import trainer_setup
    tr, _, _ = trainer_setup(dev, accum=1)
    if world == 1:      # one rank: run the collectives anyway (identities) so the RCCL path is what executes
        tr.reducer.remove()
        tr.reducer = GradReducer(tr.arenas, bucket_bytes=256 << 10, reduce_single_rank=True)
    n_coll = {"n": 0}
    orig_all_reduce = dist.all_reduce

    def counting_all_reduce(*a, **k):
        n_coll["n"] += 1
        return orig_all_reduce(*a, **k)
    dist.all_reduce = counting_all_reduce
    # the lr rule counts accumulate * world: the single-process twin uses accum 2 x world 1
    p0 = tr.arena.flat_p.clone()
    seen = {}
    orig_finish = tr.reducer.finish

    def finish():
        orig_finish()
        seen["g"] = tr.arena.flat_g.clone() / tr.scaler.scale
    tr.reducer.finish = finish
    # local (pre-exchange) gradient of this rank, for the expected mean: run the same micro-batch without exchange first
    t = torch.tensor([760, 850, 800, 720], device=dev)
    b = dict(x_start=rng.synth_input(f"dp.x{rank}", (4, 4, 32, 32), seed=48).to(dev), face_id_embs=rng.synth_input(f"dp.id{rank}", (4, 512), seed=48).to(dev),
             fg_mask=torch.ones(4, 1, 32, 32, device=dev), noise=rng.synth_input(f"dp.n{rank}", (4, 4, 32, 32), seed=48).to(dev))
    tr.optimizer.zero_grad()
    with tr.reducer.no_sync():
        loss = tr.shared_step(b, num_unet_denoising_steps=1, t=t)
        (loss * tr.scaler.scale).backward()
    local = tr.arena.flat_g.clone() / tr.scaler.scale
    tr.reducer._reset()
    tr.optimizer.zero_grad()
    tr.unet_distill_iters_count = 0
    n_coll["n"] = 0
    tr.training_step(b, 0, num_unet_denoising_steps=1, t=t)
    collectives, launch_log = n_coll["n"], list(tr.reducer.launch_log)
    tot = local.clone()
    dist.all_reduce(tot)
    if rank == 0:
        torch.save({"backend": dist.get_backend(), "collectives": collectives, "launch_log": launch_log, "flat_p": tr.arena.flat_p.cpu(), "p0": p0.cpu(), "global_step": tr.global_step, "world": tr.world, "lr": tr.learning_rate,
                    "mean_grad": seen["g"].cpu(), "mean_grad_expected_from_rank_sums": (tot / world).cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
