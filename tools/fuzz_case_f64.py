"""For one fuzz case: relative L2 distances HIP <-> f32 oracle, HIP <-> f64 oracle, f32 oracle <-> f64 oracle of every gradient.
usage: python tools/fuzz_case_f64.py <case_seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import test_gpu_fuzz as tf, test_gpu_parity as tp
orc, hr, syn = tp._mods()
c = tf._case(np.random.default_rng(1000 + int(sys.argv[1])))
print(c)
g = syn.make_gaussians(c["N"], c["W"], c["H"], seed=c["seed"], r_min=c["r_min"], r_max=c["r_max"], z_min=c["z_min"], z_max=c["z_max"])
with torch.no_grad():
    g["opacities"].mul_(c["opacity_scale"])
cam = syn.make_camera(c["W"], c["H"], pose_seed=c["pose"])
bg = torch.tensor([0.3, 0.1, 0.6])
grads = syn.make_image_grads(c["W"], c["H"], c["seed"])
f_hip, b_hip = hr.run_hip(g, cam, c["W"], c["H"], bg, grads=grads)
f32, b32 = hr.run_oracle(orc, g, cam, c["W"], c["H"], bg, grads=grads, prec="f32")
f64, b64 = hr.run_oracle(orc, g, cam, c["W"], c["H"], bg, grads=grads, prec="f64")
rl = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64).ravel() - np.asarray(b, np.float64).ravel()) / max(np.linalg.norm(np.asarray(b, np.float64).ravel()), 1e-300))
print("fragile pixels:", int((f32["fragile"] != 0).sum()), "of", f32["fragile"].size, "| mean final T", float(f_hip["final_T"].mean()))
for n in ["means3D", "means2D", "opacities", "scales", "rotations", "colors", "tau"]:
    h, o32, o64 = b_hip[n], np.asarray(b32[n]).reshape(b_hip[n].shape), np.asarray(b64[n]).reshape(b_hip[n].shape)
    print(f"{n:10s} hip-f32 {rl(h, o32):.2e} | hip-f64 {rl(h, o64):.2e} | f32-f64 {rl(o32, o64):.2e}")
for n in ["color", "depth", "opacity"]:
    print(f"{n:10s} hip-f32 {rl(f_hip[n], f32[n]):.2e} | hip-f64 {rl(f_hip[n], f64[n]):.2e} | f32-f64 {rl(f32[n], f64[n]):.2e}")
