"""N > 1 path on CPU: two gloo ranks shard independent items round-robin, reconstruct them (the
oracle stands in for the GPU compute -- this test exercises the sharding/gather plumbing, not the
kernels) and rank 0 must get exactly the unsharded result, in order."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from sr_mi355x import parallel, synth


def test_shard_indices_partition():
    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in parallel.shard_indices(n, r, world))
            assert seen == list(range(n))
            for r in range(world):
                assert all(parallel.owner_of(i, world) == r for i in parallel.shard_indices(n, r, world))
    with pytest.raises(ValueError):
        parallel.shard_indices(4, 2, 2)


def _items():
    from oracle import sr_oracle as O
    psf, shifts, f = synth.gaussian_psf(), synth.NOMINAL_4, 2
    items = []
    for s in range(5):
        truth = synth.truth_image(32, 40, seed=900 + s)
        items.append(synth.sensor_frames(np.stack([O.forward_model(truth, psf, sh, f) for sh in shifts]), seed=s))
    return items, shifts, psf, f


def _oracle_compute(lr, shifts, kernel, factor, n_iter, step):
    from oracle import sr_oracle as O
    saa = O.shift_and_add(list(lr), shifts, factor)
    return O.ibp(list(lr), shifts, kernel, saa, factor, n_iter, step)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items, shifts, psf, f = _items()
        out = parallel.reconstruct_sharded(items, shifts, psf, f, 4, 0.5, compute=_oracle_compute)
        dist.barrier()
        if rank == 0:
            q.put([(hr, errs) for hr, errs in out])
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_unsharded():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    items, shifts, psf, f = _items()
    assert len(got) == len(items)
    for lr, (hr, errs) in zip(items, got):
        hr1, errs1 = _oracle_compute(lr, shifts, psf, f, 4, 0.5)
        assert np.array_equal(hr, hr1) and errs == list(errs1)


def test_world_one_needs_no_process_group():
    items, shifts, psf, f = _items()
    out = parallel.reconstruct_sharded(items[:2], shifts, psf, f, 2, 0.5, compute=_oracle_compute)
    assert len(out) == 2 and out[0][0].shape == (32, 40)
