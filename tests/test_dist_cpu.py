"""world_size-2 gloo test of the N>1 bookkeeping bench.py uses (no GPU needed)."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "patchwork-plusplus_amd", "python"))
    import pwpp_dist
    w, r, lr = pwpp_dist.init("gloo")
    assert (w, r, lr) == (world, rank, rank)
    shard = pwpp_dist.shard_sources(6, 8, rank)
    pwpp_dist.barrier()
    elapsed, frames = pwpp_dist.aggregate(1.0 + rank, len(shard) * 3)
    per_rank = pwpp_dist.gather_values(100.0 * (rank + 1))  # every rank's own rate, in rank order
    q.put((rank, shard, elapsed, frames, per_rank))
    pwpp_dist.finalize()


def test_two_rank_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1] == [0, 1, 2, 3, 4, 5, 0, 1] and out[1][1] == [1, 2, 3, 4, 5, 0, 1, 2]
    for _, _, elapsed, frames, per_rank in out:
        assert elapsed == 2.0      # MAX over ranks
        assert frames == 48        # SUM over ranks: whole-job frames
        assert per_rank == [100.0, 200.0]  # bench.py's "per_gpu" list


def test_single_process_passthrough():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "patchwork-plusplus_amd", "python"))
    import pwpp_dist
    assert pwpp_dist.aggregate(0.5, 7) == (0.5, 7)
    assert pwpp_dist.gather_values(3.5) == [3.5]
    assert pwpp_dist.shard_sources(6, 4, 5) == [5, 0, 1, 2]
