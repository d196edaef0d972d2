#!/usr/bin/env python3
"""Diagnostics of a sequence run: the per-frame log (tracking iterations, covisibility with the last keyframe, pose error against the
ground truth), per-frame PSNR, and PNGs of rendered vs ground-truth frames.  usage: python tools/sequence_debug.py toy|half|full [outdir]"""
import os
import random
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lvdgs  # noqa: E402,F401


def write_png(path, rgb):
    """rgb: (H,W,3) uint8."""
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    chunk = lambda tag, data: struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def pose_errors(seq):
    out = []
    for i, cam in seq.cameras.items():
        est = np.eye(4); est[:3, :3] = cam.R.detach().cpu().numpy(); est[:3, 3] = cam.T.detach().cpu().numpy()
        gt = np.eye(4); gt[:3, :3] = cam.R_gt.detach().cpu().numpy(); gt[:3, 3] = cam.T_gt.detach().cpu().numpy()
        out.append((i, float(np.linalg.norm(np.linalg.inv(est)[:3, 3] - np.linalg.inv(gt)[:3, 3]))))
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "toy"
    outdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "seqdbg_" + which)
    os.makedirs(outdir, exist_ok=True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0); random.seed(0)
    import sequence_scene as ss
    from lvdgs.gaussian_renderer import render
    from lvdgs.slam_sequence import SlamSequence
    if which == "toy":
        cfg, ds, hooks, knn, psnr_only = ss.toy_sequence_on_cpu()
        seq = SlamSequence(cfg, ds.to(dev), ss.empty_map(cfg, dev), ss.PIPE, torch.zeros(3, device=dev), idle_map_iters=2).run()
    else:
        import sequence as tool
        ss = tool
        kw = dict(frames=44, scale=0.5, cadence="short", window_size=5, idle=4) if which == "half" else dict(frames=60)
        rec, seq = tool.run_sequence(dev, refine=0, **kw)
        print({k: v for k, v in rec.items() if k != "window_log"})
    for row in seq.frame_log:
        print(row)
    print("pose errors (camera centre, unaligned):", [(i, round(e, 4)) for i, e in pose_errors(seq)])
    print("map sizes:", seq.gaussian_counts)
    G = seq.frontend_gaussians
    print("opacity quantiles", torch.quantile(G.get_opacity.detach().flatten().float(), torch.tensor([0.05, 0.5, 0.95], device=dev)).tolist(),
          "max scale quantiles", torch.quantile(G.get_scaling.detach().max(1).values.float(), torch.tensor([0.05, 0.5, 0.95], device=dev)).tolist())
    from lvdgs.eval_utils import frame_metrics
    for idx in sorted(seq.cameras):
        cam = seq.cameras[idx]
        with torch.no_grad():
            pkg = render(cam, G, ss.PIPE, seq.background)
        gt = seq.dataset[idx][0]
        m = frame_metrics(pkg["render"], gt, getattr(cam, "static_mask", None), seq.background)
        print(idx, "kf" if idx in seq.kf_indices else "  ", {k: (round(v, 3) if v is not None else None) for k, v in m.items()}, "opaque", float((pkg["opacity"] > 0.95).float().mean()))
        if idx in (1, len(seq.cameras) // 2, len(seq.cameras) - 2):
            both = torch.cat([pkg["render"].clamp(0, 1), gt], 1).permute(1, 2, 0).mul(255).byte().cpu().numpy()
            write_png(os.path.join(outdir, f"frame{idx:03d}_render_over_gt.png"), both)


if __name__ == "__main__":
    main()
