"""Child process of tests/test_rccl_world1_gpu.py (not collected by pytest): the RCCL branch of the frame-sharded
render (pythoncrt_amd/shard.py; the reference's in-order commit loop crt_filter.py ref:1081-1105) on ONE GPU.

    RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=<p> python tests/_rccl_world1_child.py <height> <width> <chunk> <rounds>

A fresh process: nothing touches the GPU before init_process_group("nccl", device_id=cuda:0).  It runs, over RCCL at
world size 1,

  1. the collectives bench.py's N > 1 path uses (barrier, all_reduce MAX, all_gather, all_gather_object);
  2. ShardedRender._send_recv(send, like, src=0, dst=0): one isend + one irecv to itself inside the one
     batch_isend_irecv group, on a tensor a kernel has just written — then, WITHOUT a device synchronize in between,
     crtfx_halo_correct_batch (engine.correct) on libcrtfx's caller-supplied stream reading what RCCL produced: the hop
     is ordered behind the scan and the fix-up behind the hop by r.wait() alone;
  3. the whole protocol as the one-rank ring (ShardedRender(loopback=True)): the synchronous schedule (run_round, RCCL's
     default) and the overlapped one (submit_round / flush), parallel-hop and exact-chain rounds;

and compares every frame with the plain in-order render of the same clip on the same pipeline (FramePipeline.run).
Prints ONE JSON line; exit code 0 only when every comparison holds."""
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def diff(a, b):
    d = (a.to(torch.int16) - b.to(torch.int16)).abs()
    return int(d.max().item()), float((d != 0).float().mean().item())


def main():
    h, w, chunk, rounds = (int(x) for x in sys.argv[1:5])
    assert os.environ.get("WORLD_SIZE") == "1" and os.environ.get("RANK") == "0"
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
    torch.cuda.set_device(dev)
    res = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
           "backend_version": "rccl " + ".".join(str(x) for x in torch.cuda.nccl.version())}

    # ---- 1. the collectives of bench.py's N > 1 path ----
    dist.barrier()
    t = torch.tensor([3.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    got = [torch.zeros(1, dtype=torch.float64, device=dev)]
    dist.all_gather(got, t)
    objs = [None]
    dist.all_gather_object(objs, {"rank": 0, "x": [1, 2.5, "s"]})
    res["collectives_ok"] = bool(t.item() == 3.25 and got[0].item() == 3.25 and objs == [{"rank": 0, "x": [1, 2.5, "s"]}])
    res["init_plus_collectives_s"] = round(time.perf_counter() - t0, 2)

    from pythoncrt_amd.pipeline import FramePipeline, GpuShardEngine, baseline_config
    from pythoncrt_amd.shard import FrameShard, ShardedRender, settle_frames
    rs = baseline_config(4)[0]                   # BASELINE configs[3]: the 1080p chain with persistence 0.5 (the frame size is the caller's)
    p = rs.persistence
    n = chunk * rounds
    g = torch.Generator(device=dev).manual_seed(77)
    frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device=dev, generator=g)
    pipe = FramePipeline(dev, h, w, rs, fps=30.0, noise_seed=5)
    want, _ = pipe.run(frames, first_index=0)    # the in-order render: one state carried through the whole clip
    want = want.clone()

    # ---- 2. the self hop by hand ----
    engine = GpuShardEngine(pipe, chunk, slots=2)
    render = ShardedRender(FrameShard(1, 0, chunk), p, engine, dist=dist, loopback=True)
    x = torch.randn((h, w, 3), device=dev)       # written by a kernel on the compute stream just before the hop
    y = render._send_recv(x, x, 0, 0)
    res["self_hop_equal"] = bool(torch.equal(x, y)) and y.data_ptr() != x.data_ptr()
    local0, out0 = engine.local_scan(frames[:chunk], 0, clip_start=True, slot=0)
    local1, out1 = engine.local_scan(frames[chunk:2 * chunk], chunk, clip_start=False, slot=1)
    k = min(chunk, settle_frames(p, 2.0 ** -26))
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    carry = render._send_recv(local0[chunk - 1], local0[chunk - 1], 0, 0)      # chunk 0 starts the clip: its local final is the true one
    e1.record()
    engine.correct(local1[:k], carry, p, out1[:k])                              # no synchronize between hop and fix-up
    e2.record()
    torch.cuda.synchronize(dev)
    res["hand_hop_us"], res["hand_fixup_us"] = round(e0.elapsed_time(e1) * 1e3, 1), round(e1.elapsed_time(e2) * 1e3, 1)
    res["hand_chunk0"] = diff(out0, want[:chunk])
    res["hand_chunk1"] = diff(out1, want[chunk:2 * chunk])
    # what the fix-up had to change: the zero-state scan alone is far from the in-order frames
    raw = engine.local_scan(frames[chunk:2 * chunk], chunk, clip_start=False, slot=1)[1]
    res["uncorrected_chunk1"] = diff(raw, want[chunk:2 * chunk])

    # ---- 3. the protocol as the one-rank ring, both schedules; parallel-hop rounds and (a shorter chunk) the exact chain ----
    def ring(chunk_, overlap):
        eng = GpuShardEngine(pipe, chunk_, slots=2)
        rd = ShardedRender(FrameShard(1, 0, chunk_), p, eng, dist=dist, overlap=overlap, timing=True, loopback=True)
        outs = {}
        for r in range((n + chunk_ - 1) // chunk_):
            mine = frames[r * chunk_:(r + 1) * chunk_]
            for rr, o in rd.submit_round(mine, r, active=1):
                outs[rr] = o.clone()              # two output slots rotate: keep a copy
        for rr, o in rd.close():
            outs[rr] = o.clone()
        got_ = torch.cat([outs[i] for i in sorted(outs)])
        return {"overlap": rd.overlap, "parallel_hop": rd.parallel_hop, "diff": diff(got_, want), "schedule": rd.schedule_report()}

    res["ring_synchronous"] = ring(chunk, False)
    res["ring_overlapped"] = ring(chunk, True)
    res["ring_exact_chain"] = ring(5, False)     # p^5 > 2^-24: true finals travel, every frame of a chunk is corrected
    dist.barrier()
    dist.destroy_process_group()

    ok = res["collectives_ok"] and res["self_hop_equal"] and res["backend"] == "nccl" and res["world_size_seen"] == 1
    for key in ("hand_chunk0", "hand_chunk1"):
        ok = ok and res[key][0] <= 1 and res[key][1] < 2e-3
    for key in ("ring_synchronous", "ring_overlapped", "ring_exact_chain"):
        ok = ok and res[key]["diff"][0] <= 1 and res[key]["diff"][1] < 2e-3
    ok = ok and res["ring_synchronous"]["parallel_hop"] and res["ring_overlapped"]["overlap"] and not res["ring_exact_chain"]["parallel_hop"]
    ok = ok and res["uncorrected_chunk1"][1] > 0.01          # the comparison can fail: without the carry the frames differ
    res["ok"] = bool(ok)
    print(json.dumps(res))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
