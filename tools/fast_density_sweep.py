#!/usr/bin/env python3
"""FAST stage time of the dense kernel (ORBFE_FAST_VARIANT=1) against the sparse kernel (=0, all levels) over frames of
different corner density: KITTI-size synthetic frames with `shapes_per_mpx` random shapes (6500 = the bench's frames), 256 frames
per launch.  One process per (variant, density) because the variant is fixed when the extractor is created.
usage: python tools/fast_density_sweep.py           (drives itself through subprocesses; run on the GPU box)

Record of how the density rows of profiles/r3_fast_sparse_ab.txt were made.  The sparse kernel has since left the build
(source: profiles/r3_fast_sparse_kernel.h.txt, host side: profiles/r3_fast_sparse_host.patch): on the present library
ORBFE_FAST_VARIANT is clamped to 1..2, so the "sparse" rows would repeat the dense kernel unless that patch is applied."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

W, H, B = 1241, 376, 256


def census(img, t=7):
    """share of level-0 pixels that pass the compass test in either polarity, and with S > t"""
    I = img.astype(np.int32)
    v = I[3:-3, 3:-3]
    c0, c8, c4, c12 = I[6:, 3:-3], I[:-6, 3:-3], I[3:-3, 6:], I[3:-3, :-6]
    b = np.minimum(np.maximum(c0, c8), np.maximum(c4, c12)) > v + t
    d = np.maximum(np.minimum(c0, c8), np.minimum(c4, c12)) < v - t
    return float((b | d).mean()), float((b & d).mean())


def one(shapes, noise):
    import torch
    from orb_slam2_amd import capi, synth
    stride = (W + 63) // 64 * 64
    uniq = []
    for i in range(8):
        img = synth.textured_image(W, H, synth.BASE_SEED + 300 + i, shapes_per_mpx=shapes)
        if noise < 4:     # the generator adds +-4: rebuild the frame with less noise by smoothing the noise part away
            rng = np.random.default_rng(i)
            base = synth.textured_image(W, H, synth.BASE_SEED + 300 + i, shapes_per_mpx=shapes).astype(np.int32)
            # median of 3x3 removes +-4 speckle on flat areas but keeps the shapes' edges; then add +-noise
            pad = np.pad(base, 1, mode="edge")
            stack = np.stack([pad[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)])
            img = np.clip(np.median(stack, axis=0) + rng.integers(-noise, noise + 1, (H, W)), 0, 255).astype(np.uint8)
        uniq.append(img)
    host = np.zeros((B, H, stride), np.uint8)
    for i in range(B):
        host[i, :, :W] = uniq[i % 8]
    frames = torch.from_numpy(host).cuda()
    cap = 2100
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex = capi.ORBextractor(2000, 1.2, 8, 20, 7)
    run = lambda: ex.extract_batch_device(frames.data_ptr(), B, W, H, stride, stride * H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap)
    for _ in range(2):
        run()
    ex.set_profiling(True)
    for _ in range(6):
        run()
    p = ex.profile()
    ps, both = zip(*[census(u) for u in uniq])
    print(json.dumps({"fast_ms": round(p["stage_ms"]["fast"] / 6, 4), "variant_after": ex.fast_variant(), "compass_pass_level0": round(float(np.mean(ps)), 4),
                      "both_polarities_level0": round(float(np.mean(both)), 4), "candidates_per_frame": round(p["candidates"] / p["frames"], 1),
                      "keypoints_per_frame": round(p["keypoints"] / p["frames"], 1)}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(float(sys.argv[1]), int(sys.argv[2]))
        sys.exit(0)
    print(f"{'shapes/Mpx':>10} {'noise':>5} {'pass L0':>8} {'cand/frame':>10} {'kp/frame':>8} {'dense ms':>9} {'sparse ms':>9} {'sparse L0-2':>11}")
    for shapes, noise in ((6500, 4), (6500, 1), (2000, 4), (2000, 1), (600, 4), (600, 1), (150, 1)):
        res = {}
        for name, env in (("dense", {"ORBFE_FAST_VARIANT": "1"}), ("sparse", {"ORBFE_FAST_VARIANT": "0"}),
                          ("sparse3", {"ORBFE_FAST_VARIANT": "0", "ORBFE_FAST_SPARSE_LEVELS": "3"})):
            r = subprocess.run([sys.executable, __file__, str(shapes), str(noise)], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, **env))
            try:
                res[name] = json.loads(r.stdout.strip().splitlines()[-1])
            except (ValueError, IndexError):
                print("failed:", name, shapes, noise, (r.stdout + r.stderr)[-400:])
                sys.exit(1)
        d = res["dense"]
        flag = "" if res["sparse"]["variant_after"] == 0 else " (fell back to dense)"
        print(f"{shapes:>10} {noise:>5} {d['compass_pass_level0']:>8.3f} {d['candidates_per_frame']:>10.0f} {d['keypoints_per_frame']:>8.0f} "
              f"{d['fast_ms']:>9.4f} {res['sparse']['fast_ms']:>9.4f} {res['sparse3']['fast_ms']:>11.4f}{flag}", flush=True)
