"""Sharding of independent bases over ranks (one process per GPU) and the timing reduction used by
bench.py.  Inside one factorize nothing shards (strictly ordered pivots); across
matrices the partition is `basis b -> rank b mod world`, with no data-path collective.  The only
communication is a barrier around the timed region and a MAX over ranks of the elapsed time
(RCCL on the GPU box, gloo in the CPU tests)."""
import torch
import torch.distributed as td


def bases_of_rank(n_bases, rank, world):
    """Indices of the bases rank `rank` factorizes (round robin)."""
    return list(range(rank, n_bases, world))


def seed_of_basis(config, b):
    """Every basis of a batch is the same configuration with its own seed (C4: seeds 1..8)."""
    return int(config["seed"]) + int(b)


def fence(device=None):
    if td.is_available() and td.is_initialized():
        td.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(x, device=None):
    """MAX of a python float over all ranks (identity when not distributed)."""
    if not (td.is_available() and td.is_initialized()):
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device if device is not None else "cpu")
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, device=None):
    if not (td.is_available() and td.is_initialized()):
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device if device is not None else "cpu")
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item())


def whole_job_throughput(units_this_rank, elapsed_this_rank, device=None):
    """value = units all ranks processed / max-over-ranks elapsed time."""
    total = sum_over_ranks(units_this_rank, device)
    t = max_over_ranks(elapsed_this_rank, device)
    return total / t, t
