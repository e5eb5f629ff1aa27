"""Generates tests/golden/ref_adam_golden.npz by running the REFERENCE optimiser itself: scene/OurAdam.py is pure
Python, so its Adam class is loaded from /root/reference (read-only; by file path, to skip scene/__init__.py's
unrelated imports) and stepped on CPU with seeded float32 inputs.  Run in the build container:

    python tests/golden/make_adam_golden.py

Stored per case: initial parameters, the per-step gradients and `relevant` row lists, and the parameters and optimiser
state after the last step.  Cases: row-sparse steps (train_single.py:171-174), dense steps (relevant.size(0) == 0),
a mix, and weight decay.
"""
import importlib.util
import os

import numpy as np
import torch

REF_FILE = "/root/reference/scene/OurAdam.py"
SHAPES = dict(xyz=(3,), f_dc=(1, 3), f_rest=(15, 3), opacity=(1,), scaling=(3,), rotation=(4,))
LRS = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=5e-2, scaling=5e-3, rotation=1e-3)


def load_ref_adam():
    spec = importlib.util.spec_from_file_location("ref_ouradam", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Adam


def make_case(P, steps, mode, weight_decay, seed):
    g = torch.Generator().manual_seed(seed)
    params = {k: torch.randn(P, *s, generator=g) for k, s in SHAPES.items()}
    grads, relevants = [], []
    for it in range(steps):
        gr = {k: torch.randn(P, *s, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g)))
              for k, s in SHAPES.items()}
        if mode == "dense" or (mode == "mixed" and it % 2 == 1):
            rel = torch.empty(0, dtype=torch.int64)
        else:
            keep = torch.rand(P, generator=g) < 0.4
            gr["opacity"][~keep] = 0.0
            rel = (gr["opacity"].flatten() != 0).nonzero().flatten().long()     # train_single.py:171-172
        grads.append(gr)
        relevants.append(rel)
    return params, grads, relevants


def main():
    Adam = load_ref_adam()
    out = {}
    cases = [("sparse", 257, 5, "sparse", 0.0, 11), ("dense", 64, 4, "dense", 0.0, 12),
             ("mixed", 100, 6, "mixed", 0.0, 13), ("decay", 50, 3, "sparse", 0.01, 14)]
    out["case_names"] = np.array([c[0] for c in cases])
    for name, P, steps, mode, wd, seed in cases:
        params, grads, relevants = make_case(P, steps, mode, wd, seed)
        live = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
        opt = Adam([dict(params=[live[k]], lr=LRS[k], name=k) for k in SHAPES], lr=0.0, eps=1e-15, weight_decay=wd)
        for gr, rel in zip(grads, relevants):
            for k in SHAPES:
                live[k].grad = gr[k].clone()
            opt.step(rel)
        out[f"{name}.meta"] = np.array([P, steps, wd, seed], dtype=np.float64)
        for k in SHAPES:
            out[f"{name}.{k}.init"] = params[k].numpy()
            out[f"{name}.{k}.final"] = live[k].detach().numpy()
            st = opt.state[live[k]]
            out[f"{name}.{k}.exp_avg"] = st["exp_avg"].numpy()
            out[f"{name}.{k}.exp_avg_sq"] = st["exp_avg_sq"].numpy()
            out[f"{name}.{k}.step"] = np.array(float(st["step"]))
            for it, gr in enumerate(grads):
                out[f"{name}.{k}.grad{it}"] = gr[k].numpy()
        for it, rel in enumerate(relevants):
            out[f"{name}.relevant{it}"] = rel.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_adam_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
