"""Worker of tests/test_hip_train.py::test_data_parallel_step_two_ranks_equals_accumulated_single_process (launched with
torch.distributed.run, 2 ranks on one GPU, gloo).  Rank r trains on micro-batch r with accumulate_grad_batches=1."""
import os
import sys

import torch
import torch.distributed as dist


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    from adaface_dev_amd import rng
    from trainer_util import trainer_setup
    tr, _, _ = trainer_setup(dev, accum=1)
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
    tr.training_step(b, 0, num_unet_denoising_steps=1, t=t)
    tot = local.clone()
    dist.all_reduce(tot)
    if rank == 0:
        torch.save({"flat_p": tr.arena.flat_p.cpu(), "p0": p0.cpu(), "global_step": tr.global_step, "world": tr.world, "lr": tr.learning_rate,
                    "mean_grad": seen["g"].cpu(), "mean_grad_expected_from_rank_sums": (tot / world).cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
