"""Per-tensor error statistics of the HIP-vs-oracle comparisons, collected while the GPU tests run and written to
``gpurun_out/parity_report.json`` when the session ends (tests/conftest.py).  The summary kept under
``profiles/`` comes from that file.

For every compared tensor (a = HIP, b = oracle):
  * ``max_abs``      max |a - b|
  * ``scale``        max |b|
  * ``rel_l2``       ||a - b||_2 / ||b||_2
  * ``max_rel_sig``  max |a - b| / |b| over the elements with |b| >= 1e-3 * scale ("significant" elements: the ones
                     whose value is not itself a cancellation residue of much larger terms)
  * ``max_rel_all``  the same over every element with b != 0
  * ``frac_1e-4``    fraction of elements with |a - b| <= 1e-4 |b| (the north star's tolerance, taken literally)
"""
import json
import os

import numpy as np

REPORT = {}
_CURRENT = ["?"]


def set_test(name):
    _CURRENT[0] = name


def stats(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    if a.size == 0:
        return None
    d = np.abs(a - b)
    scale = float(np.abs(b).max())
    nb = float(np.linalg.norm(b))
    nz = b != 0
    sig = np.abs(b) >= 1e-3 * max(scale, 1e-300)
    rel = np.zeros_like(d)
    rel[nz] = d[nz] / np.abs(b[nz])
    return {
        "n": int(a.size), "max_abs": float(d.max()), "scale": scale,
        "rel_l2": float(np.linalg.norm(a - b) / nb) if nb > 0 else float(np.linalg.norm(a - b)),
        "max_rel_sig": float(rel[sig & nz].max()) if (sig & nz).any() else 0.0,
        "max_rel_all": float(rel[nz].max()) if nz.any() else 0.0,
        "frac_1e-4": float((d <= 1e-4 * np.abs(b)).mean()),
    }


def record(what, a, b):
    s = stats(a, b)
    if s is not None:
        REPORT.setdefault(_CURRENT[0], {})[what] = s
    return s


def dump(root):
    if not REPORT:
        return
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)
    # one line per tensor for people: worst cases first
    rows = [(t, w, s) for t, d in REPORT.items() for w, s in d.items()]
    rows.sort(key=lambda r: -r[2]["max_rel_sig"])
    with open(os.path.join(out, "parity_report.txt"), "w") as f:
        f.write(f"{'test':110s} {'tensor':52s} {'rel_l2':>10s} {'max_rel_sig':>12s} {'max_abs/scale':>14s} {'frac<=1e-4':>10s}\n")
        for t, w, s in rows:
            f.write(f"{t[:110]:110s} {w[:52]:52s} {s['rel_l2']:10.2e} {s['max_rel_sig']:12.2e} "
                    f"{s['max_abs'] / max(s['scale'], 1e-300):14.2e} {s['frac_1e-4']:10.4f}\n")
