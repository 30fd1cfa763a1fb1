"""r06 diagnostics: WHERE does a certified gibbs job of configs[4]'s shape leave the F32_SPLIT chain?  Replays the soak's jobs
(tools/certified_soak.py --mode gibbs --inpaint --steps 50, no direct lane) until one differs, finds the first step at which the
differing sample parts from the exact chain, and prints the report of that step for that sample: flags, gaps, the bounds in use,
the actual logit-error range and entropy errors of the rows involved."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esmdiff_amd.certified import CertifiedSampler
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.geometry import build_affine3d_from_coordinates
from esmdiff_amd.gibbs import unmask_schedule
from esmdiff_amd.weights import random_init_state_dict

MASK = 4096
sd = random_init_state_dict(cfg, seed=11, device="cuda", with_geom=True)
B, L, T = 100, 258, 50
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
exact = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
fast = Engine(cfg, sd, max_batch=B, max_len=L, precision="f16", head_precision="f32")
del sd
x0 = torch.full((B, L), MASK, dtype=torch.int64)
x0[:, 0], x0[:, -1] = 4098, 4097
x0[:, 1:-1] = torch.randint(0, 4096, (1, L - 2), generator=g)
x0[:, 97:161] = MASK
ca = torch.cumsum(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1) * 3.8, 0)
xyz = torch.stack([ca + torch.tensor([-1.2, 0.7, 0.0]), ca, ca + torch.tensor([1.3, 0.6, 0.1])], 1)
xyz[97:161] = float("inf")
xyz[0] = xyz[-1] = float("nan")
frames = build_affine3d_from_coordinates(xyz[None].repeat(B, 1, 1, 1))
sched = unmask_schedule(64, T)
table = torch.tensor(sched, dtype=torch.int32)[:, None].repeat(1, B)
cs = CertifiedSampler(fast, exact, direct_share=1.0)
cs.trace_sample = int(os.environ.get("TRACE_SAMPLE", "3"))
x0d, seqd = x0.cuda(), seq
for job in range(16):
    seed = 9000 + job
    got = cs.gibbs_sample(seq, x0, table, 1.4, 0.9, seed=seed, frames=frames)
    bounds = (cs.pair_bound(), cs.entropy_bound())
    # the exact chain, step by step, keeping every state
    exact.set_frames(*frames)
    states = [x0d.clone()]
    x = x0d.clone()
    for t, k in enumerate(sched):
        if k > 0:
            lg = exact.forward_logits(x, seqd, None)
            exact.gibbs_step(x, seqd, lg, 1.4, 0.9, torch.full((B,), k, dtype=torch.int32), seed=seed, sample_offset=0, step=t)
        states.append(x.clone())
    exact.set_frames(None)
    want = states[-1]
    bad = torch.nonzero((got != want).any(1)).flatten().tolist()
    print(f"job {seed}: differing samples {bad}; bounds P {bounds[0]:.4e} E {bounds[1]:.4e}; flagged {cs.stats['flagged']} audits {cs.stats['audit_checked']}", flush=True)
    if not bad:
        continue
    s = bad[0]
    # first step at which the certified result disagrees with what the exact chain unmasked there
    t_star = None
    for t in range(T):
        newly = (states[t][s] == MASK) & (states[t + 1][s] != MASK)
        if bool((got[s][newly] != states[t + 1][s][newly]).any()):
            t_star = t
            break
    print(f"  sample {s} parts from the exact chain at step {t_star} (k = {sched[t_star]})")
    for ev in cs.stats.get("trace", []):
        if abs(ev[1] - t_star) <= 2:
            print("   trace", ev)
    xs, sq = states[t_star][s:s + 1].clone(), seqd[s:s + 1]
    fr1 = tuple(f[s:s + 1] for f in frames)
    for eng in (fast, exact):
        eng.set_frames(*fr1)
    lg_f = fast.forward_logits(xs, sq, None).clone()
    lg_e = exact.forward_logits(xs, sq, None).clone()
    for eng in (fast, exact):
        eng.set_frames(None)
    par = torch.from_numpy(fast.gibbs_step_params_host(np.array([s]), np.array([t_star]), np.array([sched[t_star]]))).cuda()
    flags = torch.zeros(1, dtype=torch.int32, device="cuda")
    gaps = torch.full((1, 2), float("inf"), device="cuda")
    xf = fast.gibbs_step_rows(xs.clone(), sq, lg_f, 1.4, 0.9, par, seed=seed, pair_bound=bounds[0], entropy_bound=bounds[1], flags=flags, gaps=gaps)
    xe = exact.gibbs_step_rows(xs.clone(), sq, lg_e, 1.4, 0.9, par, seed=seed)
    print(f"  replay of that step alone: fast flags {int(flags[0])} gaps {gaps[0].tolist()}; fast == exact: {bool(torch.equal(xf, xe))}; "
          f"exact replay == chain: {bool(torch.equal(xe[0], states[t_star + 1][s]))}")
    st = exact.logit_error_stats(lg_f, lg_e, xs, True)[0].cpu().numpy()
    m = (xs[0] == MASK).cpu().numpy()
    Hf = -(torch.log_softmax(lg_f[0].double(), -1).exp() * torch.log_softmax(lg_f[0].double(), -1)).sum(-1).cpu().numpy()
    He = -(torch.log_softmax(lg_e[0].double(), -1).exp() * torch.log_softmax(lg_e[0].double(), -1)).sum(-1).cpu().numpy()
    rows = np.nonzero(m)[0]
    of, oe = rows[np.argsort(Hf[rows])], rows[np.argsort(He[rows])]
    print(f"  masked rows {len(rows)}; largest range {st[m, 4].max():.3e} (P {bounds[0]:.3e}); entropy errors max {np.abs(st[m, 5]).max():.3e} "
          f"(2E {2 * bounds[1]:.3e}); fast order {of[:4].tolist()} H {Hf[of[:4]].round(7).tolist()}; exact order {oe[:4].tolist()} H {He[oe[:4]].round(7).tolist()}")
    print(f"  fast unmasked {torch.nonzero(xf[0] != xs[0]).flatten().tolist()} -> {xf[0][xf[0] != xs[0]].tolist()}; exact unmasked "
          f"{torch.nonzero(xe[0] != xs[0]).flatten().tolist()} -> {xe[0][xe[0] != xs[0]].tolist()}")
    break
