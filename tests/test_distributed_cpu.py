"""N>1 path on CPU (gloo, world_size 2): the sharding + gather + merge logic bench.py uses for the sharded kNN-2 and the
frame-sharding arithmetic for batched extraction.  The per-shard scan is played by the oracle here (no GPU in this
container); on the GPU box tests/test_gpu_match.py::test_knn2_sharded_merge_equals_unsharded runs the same merge on
kernel outputs."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    from orb_slam2_amd import capi, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    NQ, NDB = 64, 6000
    db = synth.random_descriptors(NDB, 11)
    db[4000] = db[5]                                   # same row in both shards: lowest global index must win
    shard = NDB // world
    q = torch.from_numpy(synth.planted_queries(db, NQ, 12, planted_frac=0.5)) if rank == 0 else torch.zeros((NQ, 32), dtype=torch.uint8)
    dist.broadcast(q, src=0)                            # C2: query broadcast
    qn = q.numpy()
    if rank == 0:
        qn[0] = db[5]
    dist.broadcast(q, src=0)
    bd, bi, sd = O.knn2(q.numpy(), db[rank * shard:(rank + 1) * shard])
    res = torch.from_numpy(np.stack([bd, np.where(bi >= 0, bi + rank * shard, -1), sd]).astype(np.int32))
    gathered = [torch.zeros_like(res) for _ in range(world)]
    dist.all_gather(gathered, res)                      # C1: 12 bytes per query and shard
    g = torch.stack(gathered).numpy()
    merged = capi.merge_knn2(g[:, 0], g[:, 1], g[:, 2])
    ref = O.knn2(q.numpy(), db)
    ok = all(np.array_equal(a, b) for a, b in zip(merged, ref))
    # frame sharding of config 4: contiguous blocks of ceil(4541/G) frames, every frame exactly once
    total = 4541
    per = (total + world - 1) // world
    mine = torch.tensor([max(0, min(total, (rank + 1) * per) - rank * per)], dtype=torch.int64)
    dist.all_reduce(mine)
    ok = ok and int(mine.item()) == total
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        Path(out_path).write_text(str(int(flag.item())))
    dist.destroy_process_group()


def test_sharded_knn2_gloo_world2(tmp_path):
    out = tmp_path / "ok.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text() == "1"


def test_bench_frame_plan_covers_the_sequence_once():
    """bench.py's split of the 4541-frame step (BASELINE.json configs[3]) over 1 / 2 / 4 / 8 ranks and further counts: contiguous
    blocks that cover every frame exactly once, equal launches of at most --batch frames, no small ragged tail launch."""
    sys.path.insert(0, str(ROOT))
    import bench
    for total in (4541, 1, 7, 640, 641, 5000):
        for world in (1, 2, 3, 4, 8, 16):
            nxt = 0
            for rank in range(world):
                first, B, launch = bench.plan_rank(total, world, rank, 640)
                assert first == min(nxt, total) and 0 <= B and 1 <= launch <= 640
                nxt = first + B
                if B:
                    sizes = [min(launch, B - f0) for f0 in range(0, B, launch)]
                    assert sum(sizes) == B and len(sizes) == -(-B // 640)          # as few launches as the cap allows
                    assert max(sizes) - min(sizes) < len(sizes)                    # equal up to the rounding of the division
            assert nxt == total
    assert bench.plan_rank(4541, 1, 0, 640) == (0, 4541, 568)
    assert bench.plan_rank(4541, 8, 7, 640) == (3976, 565, 565)
    assert bench.plan_rank(0, 4, 2, 512) == (1024, 512, 512)                       # weak scaling: --batch frames per rank
