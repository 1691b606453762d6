#!/usr/bin/env python3
"""Randomised stress of the exact row-lazy Adam (run on a GPU box; collected through tests/test_gpu_fuzz.py):
    python tests/fuzz_rowlazy.py --cases 200 --seed 1
Every case draws sizes, tensor shapes ([N, w] / [N, T, w], w in 1 .. 70), hyper-parameters (incl. weight decay, where the
caught rows must be ignored) and a random sequence of frames -- peek + step with the caught rows, peek without a step (an
evaluation frame), in-place catch-up + step, step without any catch-up, LIST or SCAN form, learning-rate changes, flush(),
state_dict() round trips into a fresh optimizer -- and checks BIT-IDENTITY with the optimizer that steps every row every time:
the peeked rows at every frame, parameters and both moments after every flush and at the end."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd.optim import FusedAdam  # noqa: E402


FLUSH_EVERY = False


def case(seed, dev):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    N = ri(1, 4000)
    n_t = ri(1, 3)
    shapes = []
    for _ in range(n_t):
        T = ri(1, 4) if ri(0, 1) else 1
        w = ri(1, 70) if ri(0, 3) == 0 else [3, 45, 48, 1, 16][ri(0, 4)]
        shapes.append((T, w))
    wd = 0.0 if ri(0, 3) else 1e-2
    eps = [1e-15, 1e-8][ri(0, 1)]
    base = [torch.randn(N, T, w, generator=g) * 0.3 if T > 1 else torch.randn(N, w, generator=g) * 0.3 for T, w in shapes]
    dense_extra = torch.randn(N, 3, generator=g)

    def make(lazy):
        P = [b.clone().to(dev).requires_grad_(True) for b in base]
        E = dense_extra.clone().to(dev).requires_grad_(True)
        o = FusedAdam([{"params": P, "lr": 1e-2}, {"params": [E], "lr": 1e-3}], eps=eps, weight_decay=wd)
        if lazy:
            for p, (T, w) in zip(P, shapes):
                o.set_row_lazy(p, traversals=T if T > 1 else None, hist_capacity=ri(2, 40))
        return P, E, o
    Pa, Ea, oa = make(False)
    Pb, Eb, ob = make(True)
    stride = sum(w for _, w in shapes) + ri(0, 5)
    cols, c = [], 0
    for _, w in shapes:
        cols.append(c)
        c += w

    hist_log = []

    def check(tag):
        ob.flush()
        for k, (pa, pb) in enumerate(zip(Pa, Pb)):
            if not torch.equal(pa, pb):
                d = (pa != pb).reshape(N, -1)
                mm = float((oa.state[pa]["exp_avg"] - ob.state[pb]["exp_avg"]).abs().max())
                vv = float((oa.state[pa]["exp_avg_sq"] - ob.state[pb]["exp_avg_sq"]).abs().max())
                raise AssertionError((seed, tag, k, "p", dict(N=N, shapes=shapes, wd=wd, eps=eps, m_diff=mm, v_diff=vv,
                                                              lr=[g_["lr"] for g_ in ob.param_groups], lr_a=[g_["lr"] for g_ in oa.param_groups],
                                                              hyper=ob._hyper_dev.tolist(), hyper_a=oa._hyper_dev.tolist(),
                                                              n_diff_rows=int(d.any(1).sum()),
                                                              rows=d.any(1).nonzero().flatten()[:10].tolist(), cols=d.any(0).nonzero().flatten()[:10].tolist(),
                                                              max=float((pa - pb).abs().max()), history=hist_log)))
            if pa in oa.state:
                assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]), (seed, tag, k, "m")
                assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]), (seed, tag, k, "v")
        assert torch.equal(Ea, Eb), (seed, tag, "dense")

    steps = ri(3, 40)
    for s in range(steps):
        frac = [0.0, 0.02, 0.2, 0.6, 1.0][ri(0, 4)]
        vis = torch.rand(N, generator=g) < frac
        ids = torch.nonzero(vis).reshape(-1).to(torch.int32)
        n_vis = ids.numel()
        cap = n_vis + ri(0, 9)
        row_of = torch.full((N,), -1, dtype=torch.int32)
        row_of[vis] = torch.arange(n_vis, dtype=torch.int32)
        row_of = row_of.to(dev)
        ids_d = torch.cat([ids, torch.full((cap - n_vis,), 2 ** 30, dtype=torch.int32)]).to(dev) if cap else torch.zeros(1, dtype=torch.int32, device=dev)
        totals = torch.tensor([(n_vis << 32) | 3], dtype=torch.int64, device=dev)
        rows = (torch.randn(max(cap, 1), stride, generator=g) * 0.02).to(dev)
        ts = [ri(0, T - 1) if T > 1 else None for T, _ in shapes]
        lst = bool(ri(0, 1)) and cap > 0
        rid = ((ids_d, 0, totals if ri(0, 1) else None),) if lst else ()
        if ri(0, 4) == 0:
            new_lr = 1e-2 * (0.5 + float(torch.rand(1, generator=g)))
            for o in (oa, ob):
                o.param_groups[0]["lr"] = new_lr
        mode = ri(0, 4)
        hist_log.append((s, mode, lst, n_vis, ts))
        # 0, 1: peek + caught step; 2: peek only (evaluation); 3: in-place catch-up + step; 4: step alone
        C = None
        if mode in (0, 1, 2) and cap > 0:
            C = torch.full((cap, stride), float("nan"), device=dev)
            ob.peek_rows([(p, row_of, t, col) + rid for p, t, col in zip(Pb, ts, cols)], C)
            r = row_of[vis.to(dev)].long()
            for pa, t, col, (T, w) in zip(Pa, ts, cols, shapes):
                want = pa[vis.to(dev)] if t is None else pa[vis.to(dev), t]
                got = C[r, col:col + w]
                if not torch.equal(got, want.reshape(-1, w)):
                    bad = (got != want.reshape(-1, w)).nonzero()
                    raise AssertionError((seed, s, "peek", dict(N=N, shapes=shapes, wd=wd, eps=eps, lst=lst, cap=cap, n_vis=n_vis, ts=ts,
                                                                 col=col, w=w, n_bad=int(bad.shape[0]), first=bad[:3].tolist(),
                                                                 got=got[bad[0, 0], bad[0, 1]].item(), want=want.reshape(-1, w)[bad[0, 0], bad[0, 1]].item(),
                                                                 history=hist_log)))
        elif mode == 3:
            ob.catch_up_rows([(p, row_of, t) + rid for p, t in zip(Pb, ts)])
        if mode == 2:
            continue
        ge = (torch.randn(N, 3, generator=g) * 0.1).to(dev)
        for P, E, o, lazy in ((Pa, Ea, oa, False), (Pb, Eb, ob, True)):
            E.grad = ge.clone()
            for p, t, col in zip(P, ts, cols):
                kw = {}
                if lazy and C is not None and mode in (0, 1):
                    kw["caught"] = (C, col)
                if lazy and lst:
                    kw["row_ids"] = rid[0]
                o.set_row_gradient(p, rows, row_of, col, slice_index=t, **kw)
            o.step()
        if ri(0, 6) == 0 or FLUSH_EVERY:
            check(("flush", s))
        if ri(0, 9) == 0:          # a state_dict round trip into a fresh lazy optimizer
            sd = ob.state_dict()
            P2, E2, o2 = make(True)
            with torch.no_grad():
                for q, p in zip(P2 + [E2], Pb + [Eb]):
                    q.copy_(p)
            o2.load_state_dict(sd)
            for grp_new, grp_old in zip(o2.param_groups, ob.param_groups):
                grp_new["lr"] = grp_old["lr"]
            Pb, Eb, ob = P2, E2, o2
    check("end")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--flush-every-step", action="store_true")
    a = ap.parse_args()
    global FLUSH_EVERY
    FLUSH_EVERY = a.flush_every_step
    dev = torch.device("cuda")
    for i in range(a.cases):
        case(a.seed * 100003 + i, dev)
    torch.cuda.synchronize()
    print(f"row-lazy fuzz ok: {a.cases} cases from seed {a.seed}")


if __name__ == "__main__":
    main()
