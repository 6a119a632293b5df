#!/usr/bin/env python3
"""bench.py -- HMC leapfrog chain-steps/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): FFHQ 256x256x3, deg = inpaint_random (M = 15 729), sigma_0 = 0.05,
tau = 1.0, eps = 0.05, timesteps = 3, 64 chains PER GPU (weak scaling: chains are sharded, never data),
synthetic inputs, guided-diffusion FFHQ U-Net architecture with random-init fp32 weights (the checkpoint
is fetch-only; random init is also the reference's own fallback).

A "step" is one leapfrog step of every chain of the batch, exactly as the sampler executes it
(`LeapfrogEngine.step`): the decode (3 x [score forward + DDIM mix]), the data term, the backward
(3 x [DDIM mix VJP + score input-gradient]) and the fused momentum+position update -- nothing is skipped or
cached.  value = N * 64 * K / (max-over-ranks time).

Extra objects on the same line:
  roofline      the dominant HIP kernel (fused leapfrog update, 20 B/element = 5T per chain):
                algorithmic bytes per launch / average launch duration, HIP events on the launch stream
                over a region of back-to-back launches that rotates over buffer sets larger than the
                256 MiB Infinity Cache, so every launch streams from HBM as it does between two score
                evaluations.  `in_situ_us` is the same kernel timed per launch inside the timed steps.
                `traffic` is the PMC figure of the committed profile named in `traffic_source` (counters need a
                rocprofv3 pass of their own; they cannot be read inside this run).
  hot_path_only the HIP side of one step THROUGH THE ENGINE (the score replaced by a resident tensor whose
                backward hands back a resident gradient): what the sampler launches, minus the U-Net.
  by_deg        configs[2] / configs[3] operators (sr4, deblur_aniso): hot path, dominant data-term kernel
                against its roofline (fp32 MFMA for the spectral chain), end-to-end chain-steps/s of a few steps;
                configs[4] (hmc_latent, 16 chains): a few end-to-end steps (`--latent` prints its full line).
  cpu_baseline  the oracle (CPU restatement, validated bit-exact against the reference) on the host cores:
                with the U-Net in the loop (B = 1) and score-stubbed at B = 1 and B = 64.

`--latent` prints the line of BASELINE configs[4] instead (hmc_latent: 16 chains per GPU of [3,64,64] latents, LDM
U-Net + VQ-f4 first stage at the configs/config_ffhq_latent.yml widths, random init, inpaint_random at 256x256).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
B_PER_GPU, DIM, CH = 64, 256, 3
B_LATENT, ZDIM = 16, 64          # configs[4]: batch 128 over 8 GPUs
SIGMA0_CLI, TAU, EPS, TIMESTEPS = 0.05, 1.0, 0.05, 3
SPECTRAL_FLOP_PER_CHAIN = 805.3e6          # 8 GEMMs x 3 channels x 33.55 MFLOP


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--chunk', type=int, default=None,
                    help='chains per score-network call (activation memory); default: the whole batch -- with the fused GroupNorm '
                         'kernels the three autograd graphs of 64 chains fit the 288 GB (round 1 needed 32)')
    ap.add_argument('--batch', type=int, default=None, help='chains per GPU (BASELINE: 64; 16 with --latent)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-by-deg', action='store_true', help='skip the sr4 / deblur_aniso legs')
    ap.add_argument('--no-full-run', action='store_true', help='skip the whole-trajectory leg (sampler.hmc_chains itself)')
    ap.add_argument('--full-run-tau', type=float, default=0.25,
                    help='tau of the whole-trajectory leg (L = floor(tau / 0.05) leapfrog steps per trajectory; the reference '
                         'default 1.0 gives L = 20 = 33 s per trajectory at 64 chains)')
    ap.add_argument('--full-run-trajectories', type=int, default=4)
    ap.add_argument('--kernel-only', action='store_true', help='skip the end-to-end steps (profiling the HIP kernels)')
    ap.add_argument('--roofline-launches', type=int, default=200)
    ap.add_argument('--deg', default='inpaint_random',
                    help='degradation: inpaint_random (the BASELINE metric), sr4 (configs[2]), deblur_aniso (configs[3]), ...')
    ap.add_argument('--latent', action='store_true', help='BASELINE configs[4]: hmc_latent with the LDM U-Net + VQ-f4 decode in the loop')
    ap.add_argument('--tiny-score', action='store_true',
                    help='rehearsal only: a 32-channel U-Net of the same architecture (control-flow tests; never the metric)')
    ap.add_argument('--rehearse-shared-gpu', action='store_true',
                    help='rehearsal only: all ranks use cuda:0 over gloo (checks the N>1 control flow on a 1-GPU box)')
    return ap.parse_args()


def build_problem(device, B, chain_id0, seed=5678, deg='inpaint_random', model=None, tiny=False, sigma0=SIGMA0_CLI):
    import nhmc.kernels as K
    from nhmc import operators, plugin, schedule, unet
    gen = torch.Generator().manual_seed(seed)
    op = operators.build_operator(deg, CH, DIM, device, generator=gen)
    if model is None:
        torch.manual_seed(seed)
        cfg = dict(unet.FFHQ_CONFIG, num_channels=32, num_head_channels=32) if tiny else unet.FFHQ_CONFIG
        model = unet.create_model(**cfg).to(device).eval().requires_grad_(False)
    algo = plugin.HMC(model, op, 2 * sigma0)
    b = torch.from_numpy(schedule.get_beta_schedule('linear', beta_start=1e-4, beta_end=0.02,
                                                    num_diffusion_timesteps=1000)).float().to(device)
    seq, seq_next = schedule.timestep_ladder(1000, TIMESTEPS)
    shape = (B, CH, DIM, DIM)
    x = K.randn_philox(shape, seed, chain_id0, 0, device=device)
    p = K.randn_philox(shape, seed, chain_id0, 1, device=device)
    x_true = K.randn_philox(shape, seed, chain_id0, 2, device=device).clamp_(-1, 1)
    y = op.H(x_true) + (2 * sigma0) * torch.randn(B, op.M, device=device,
                                                  generator=torch.Generator(device=device).manual_seed(seed + chain_id0))
    return dict(op=op, algo=algo, b=b, seq=seq, seq_next=seq_next, x=x, p=p, y=y, model=model, sigma0=sigma0, deg=deg)


def leapfrog_roofline(device, B, launches, n_elem=CH * DIM * DIM):
    """Dominant kernel: nhmc_leapfrog_fused(MID).  Rotates over R buffer sets (x,p,g) whose total
    footprint exceeds the Infinity Cache, events around the whole region on the launch stream."""
    import nhmc.kernels as K
    N = n_elem
    per_set = 3 * B * N * 4
    R = max(2, math.ceil(3 * (256 << 20) / per_set))                 # >= 3x the 256 MiB Infinity Cache
    sets = [tuple(K.randn_philox((B, N), 1, 0, 3 * r + k, device=device) for k in range(3)) for r in range(R)]
    eps = torch.full((B,), 1e-3, dtype=torch.float64, device=device)
    sig = torch.full((B,), 1.7, dtype=torch.float64, device=device)
    for i in range(max(R, min(100, launches))):                      # warm-up (code object load, TLB, the clock ramp)
        s = sets[i % R]
        K.leapfrog_fused(K.LF_MID, s[0], s[1], s[2], eps, sig, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(launches):
        s = sets[i % R]
        K.leapfrog_fused(K.LF_MID, s[0], s[1], s[2], eps, sig, 1.0)
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) * 1e-3 / launches
    alg_bytes = 5 * B * N * 4                                        # R x,p,g + W x,p  (20 B/element)
    # measured copy-kernel bandwidth: a plain streaming copy with the same access pattern, same buffers
    flat = [t for s in sets for t in s]
    for i in range(len(flat)):
        K.copy_probe(flat[i], flat[(i + 1) % len(flat)])
    torch.cuda.synchronize()
    e0.record()
    for i in range(launches):
        K.copy_probe(flat[i % len(flat)], flat[(i + 4) % len(flat)])
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 2 * B * N * 4 / (e0.elapsed_time(e1) * 1e-3 / launches) / 1e9
    del sets, flat
    torch.cuda.empty_cache()
    return dict(kernel='k_leapfrog<MID> (nhmc_leapfrog_fused)', avg_us=avg_s * 1e6, bytes_per_launch=alg_bytes,
                achieved=alg_bytes / avg_s / 1e9, launches=launches, chains_per_launch=B, buffer_sets=R,
                footprint_mib=R * per_set / 2 ** 20, copy_kernel_gbs=copy_gbs,
                frac_of_copy_kernel=alg_bytes / avg_s / 1e9 / copy_gbs,
                copy_note='nhmc_copy_probe: 1 read : 1 write streaming copy, same buffers; the update is 3 reads : 2 writes')


def kernel_table(device, prob, B, launches=40):
    """Every HIP kernel of one inpaint_random leapfrog step, timed alone with HIP events at the sampler's launch size
    (B chains), rotating over buffer sets > the Infinity Cache: average launch time, algorithmic bytes,
    fraction of the 8 TB/s peak.  The same kernels appear in profiles/rNN_kernel_stats_hip.csv (rocprofv3)."""
    import nhmc.kernels as K
    op = prob['op']
    N = CH * DIM * DIM
    T = N * 4 * B
    R = 3
    sets = [dict(x=K.randn_philox((B, CH, DIM, DIM), 11, 0, 5 * r, device=device),
                 e=K.randn_philox((B, 2 * CH, DIM, DIM), 11, 0, 5 * r + 1, device=device),
                 g=K.randn_philox((B, CH, DIM, DIM), 11, 0, 5 * r + 2, device=device),
                 g2=K.randn_philox((B, CH, DIM, DIM), 11, 0, 5 * r + 3, device=device),
                 p=K.randn_philox((B, CH, DIM, DIM), 11, 0, 5 * r + 4, device=device),
                 ge=torch.zeros(B, 2 * CH, DIM, DIM, device=device)) for r in range(R)]
    at = torch.full((B,), 0.0777966604, device=device)
    an = torch.full((B,), 0.5214230418, device=device)
    one = torch.ones(B, device=device)
    eps = torch.full((B,), 1e-3, dtype=torch.float64, device=device)
    sig = torch.full((B,), 1.7, dtype=torch.float64, device=device)
    y = prob['y']
    cases = [
        ('k_mix_fwd (nhmc_ddim_mix_fwd)', 3 * T, lambda s: K.ddim_mix_fwd(s['x'], s['e'], at, an)),
        ('k_mix_bwd, two upstream gradients (nhmc_ddim_mix_bwd)', 6 * T,
         lambda s: K.ddim_mix_bwd(s['g'], s['x'], s['e'], at, an, gout2=s['g2'], g_e_out=s['ge'])),
        ('data term + last-step VJP fused, as a MID step launches it (operator.fused_last_vjp, loss partials not summed)',
         4 * T + int(op.M) * 4 * B, lambda s: op.fused_last_vjp(s['x'], s['e'], an, one, y, g_e_out=s['ge'], loss_out=K.NO_LOSS)),
        ('the same with the per-chain loss (first / last step of a trajectory: + the partial-sum kernel)', 4 * T + int(op.M) * 4 * B,
         lambda s: op.fused_last_vjp(s['x'], s['e'], an, one, y, g_e_out=s['ge'])),
        ('k_leapfrog<MID>, second gradient pointer (nhmc_leapfrog_fused)', 6 * T,
         lambda s: K.leapfrog_fused(K.LF_MID, s['x'], s['p'], s['g'], eps, sig, 1.0, g2=s['g2'])),
    ]
    rows = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, nbytes, fn in cases:
        for r in range(R):
            fn(sets[r])
        torch.cuda.synchronize()
        e0.record()
        for i in range(launches):
            fn(sets[i % R])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / launches
        rows.append(dict(kernel=name, avg_us=round(us, 2), bytes_per_launch=nbytes, gbs=round(nbytes / us / 1e3, 1),
                         frac=round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)))
    del sets
    torch.cuda.empty_cache()
    return rows


class _ResidentScore(torch.autograd.Function):
    """Score stand-in for the hot-path leg: forward hands out a resident [B,2C,H,W] tensor, backward a resident
    input-gradient -- no arithmetic, so what is timed is exactly what the engine launches around the U-Net."""

    @staticmethod
    def forward(ctx, x, e, gs):
        ctx.gs = gs
        return e.view_as(e)

    @staticmethod
    def backward(ctx, g_e):
        return ctx.gs, None, None


def hot_path_only(device, prob, B, steps, chunk=None):
    """The HIP side of one leapfrog step through LeapfrogEngine.step (kernel-comparable number)."""
    import nhmc.kernels as K
    from nhmc import sampler
    e = K.randn_philox((B, 2 * CH, DIM, DIM), 7, 0, 0, device=device)
    gs = K.randn_philox((B, CH, DIM, DIM), 7, 0, 1, scale=1e-3, device=device)

    def score(x, t):
        n = x.shape[0]
        return _ResidentScore.apply(x, e[:n], gs[:n])

    eng = sampler.LeapfrogEngine(score, prob['op'], prob['b'], prob['seq'], prob['seq_next'], device, chunk=chunk)
    x, p, y = prob['x'].clone(), prob['p'].clone(), prob['y']
    eps = torch.full((B,), EPS, dtype=torch.float64, device=device)
    sig = torch.full((B,), 1.7, dtype=torch.float64, device=device)
    ws = K.leapfrog_ws(B, x[0].numel(), device)
    for _ in range(3):
        eng.step(K.LF_MID, x, x, p, y, eps, sig, 1.0, ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(K.LF_MID, x, x, p, y, eps, sig, 1.0, ws)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(value=B * steps / dt, unit='chain-steps/s', ms_per_step=1e3 * dt / steps, steps=steps,
                note='LeapfrogEngine.step with a resident score (no U-Net): 3 mix fwd + (data term + last VJP fused) + 2 mix VJP '
                     '+ fused update, as the sampler launches them')


def data_term_roofline(device, prob, B, launches=100, warm=100):
    """Dominant data-term kernel of a non-inpaint operator: the fused (data term + last VJP) call, timed with events.
    deblur_aniso: 8 fp32-MFMA products, 805.3 MFLOP per chain against the fp32 matrix peak; sr4: 4T + y against HBM."""
    import nhmc.kernels as K
    op = prob['op']
    e = K.randn_philox((B, 2 * CH, DIM, DIM), 9, 0, 0, device=device)
    ge = torch.zeros_like(e)
    at = torch.full((B,), 0.5214230418, device=device)
    atn = torch.ones(B, device=device)
    x, y = prob['x'], prob['y']
    xt_next = K.ddim_mix_fwd(x, e, at, atn, final_clip=True)['xt_next']
    extra = dict(xt_next=xt_next) if getattr(op, 'fused_wants_decode', False) else {}

    def call():                                                # as a MID leapfrog step launches it: gradient only, no loss summation
        return op.fused_last_vjp(x, e, at, atn, y, g_e_out=ge, loss_out=K.NO_LOSS, **extra)
    # 100 untimed calls first: after 3 (round 2 and the first r3 runs) the chip is still on its clock ramp and the MFMA-bound
    # chain reads 7-12 % slower than in a run that keeps the GPU busy (tools/pair_bench.py: 506 vs 454 us per data term)
    for _ in range(warm):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        call()
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) * 1e-3 / launches

    def measured_traffic(key):
        """HBM bytes per call from the committed PMC pass (profiles/traffic_ops.json), valid at the batch it was taken at."""
        tpath = os.path.join(ROOT, 'profiles', 'traffic_ops.json')
        if not os.path.exists(tpath):
            return {'traffic': None}
        with open(tpath) as f:
            rec = json.load(f)
        if rec.get('chains_per_launch') != B or key not in rec:
            return {'traffic': None}
        return {'traffic': rec[key]['hbm_bytes_per_call'], 'traffic_source': rec.get('source')}
    if hasattr(op, 'factors'):
        proj = bool(getattr(op, 'projected', False))                   # 4 products (residual in the left singular basis) instead of 8
        flops = SPECTRAL_FLOP_PER_CHAIN * B // (2 if proj else 1)
        return dict(bound='mfma', achieved=round(flops / avg_s / 1e12, 1), peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                    frac=round(flops / avg_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), avg_us=round(avg_s * 1e6, 1),
                    **measured_traffic('deblur_aniso_projected' if proj else 'deblur_aniso'),
                    kernel='spectral chain: data term + last-step VJP (%s), fp32 MFMA, %d products'
                           % ('nhmc_data_spectral_proj_vjp' if proj else 'nhmc_data_spectral_vjp', 4 if proj else 8),
                    flops_per_call=flops, dtype='f32')
    T = CH * DIM * DIM * 4
    alg = (4 * T + op.M * 4) * B
    return dict(bound='hbm', achieved=round(alg / avg_s / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                frac=round(alg / avg_s / 1e9 / HBM_PEAK_GBS, 4), avg_us=round(avg_s * 1e6, 1),
                **measured_traffic(prob.get('deg', '')),
                kernel='data term fused with the last-step VJP (R xt, e; W g_xt, g_e; R y)', bytes_per_call=alg)


def host_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 CPUs but grants a 16-CPU quota; oversubscribing it stalls the oracle)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(seed=5678):
    """Oracle leapfrog steps on the host cores: (i) with the FFHQ U-Net in the loop at B = 1 -- `value`,
    the number comparable with the headline metric; (ii) score-stubbed at B = 1 and B = 64 -- comparable with
    `hot_path_only` and the HIP kernels."""
    from oracle import hmc_ref, operators as oops, schedule as osched
    from nhmc import unet
    torch.set_num_threads(host_cores())
    gen = torch.Generator().manual_seed(seed)
    missing = oops.random_inpaint_missing(DIM, generator=gen)
    op = oops.InpaintRef(CH, DIM, missing)
    torch.manual_seed(seed)
    net = unet.create_model(**unet.FFHQ_CONFIG).eval().requires_grad_(False)
    b = osched.betas_fp32()
    seq, seq_next = osched.timestep_ladder(1000, TIMESTEPS)

    def run(model, B, steps):
        x = torch.randn(B, CH, DIM, DIM, generator=gen)
        p = torch.randn(B, CH, DIM, DIM, generator=gen)
        y = op.H(torch.rand(B, CH, DIM, DIM, generator=gen) * 2 - 1) + 0.1 * torch.randn(B, op.M, generator=gen)
        t0 = time.perf_counter()
        for _ in range(steps):
            xl = x.clone().requires_grad_(True)
            _, _, _, g = hmc_ref._data_loss_and_grad(xl, b, seq, seq_next, model, op, y)      # decode + gradient
            x, p = hmc_ref.leapfrog_update('mid', x, p, g, eps=EPS, sigma_y=1.7, m=1.0)[:2]   # momentum + position
        return time.perf_counter() - t0

    def stub(x, t):                        # two elementwise ops standing in for the score (differentiable)
        return torch.cat([0.5 * x, 0.25 * x], dim=1)

    steps = 4                                                                       # ~15-20 s on the box's 16-CPU quota
    dt = run(net, 1, steps)
    run(stub, 1, 2)
    s1, n1 = run(stub, 1, 20), 20
    s64, n64 = run(stub, 64, 3), 3
    return dict(value=steps / dt, unit='chain-steps/s', cores=torch.get_num_threads(), kind='port',
                sample=f'oracle (oracle/hmc_ref.py) leapfrog steps, B=1, {steps} consecutive steps, FFHQ U-Net fp32 on CPU: {dt:.1f} s',
                score_stubbed=dict(b1_chain_steps_per_s=round(n1 / s1, 1), b1_ms_per_step=round(1e3 * s1 / n1, 2),
                                   b64_chain_steps_per_s=round(64 * n64 / s64, 1), b64_ms_per_step=round(1e3 * s64 / n64, 1),
                                   sample=f'same oracle step with a two-op stand-in score: {n1} steps at B=1 ({s1:.2f} s), '
                                          f'{n64} steps at B=64 ({s64:.1f} s); compare with hot_path_only'))


def single_chain_rate(eng, x, p, y, eps, sig, with_graph):
    """The reference's own operating point: ONE chain (its hmc() is batch-1 only); BASELINE.md derives ~3.2 leapfrog
    steps/s for it from the authors' logs (unstated NVIDIA GPU).  Runs last: a failed graph capture must not disturb
    the measurements above."""
    import nhmc.kernels as K
    x1, p1, y1 = x[:1].clone(), p[:1].clone(), y[:1].contiguous()
    e1, s1 = eps[:1].contiguous(), sig[:1].contiguous()
    ws = K.leapfrog_ws(1, x1[0].numel(), x1.device)

    def rate(graph):
        for _ in range(2):
            eng.step(K.LF_MID, x1, x1, p1, y1, e1, s1, 1.0, ws, graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.step(K.LF_MID, x1, x1, p1, y1, e1, s1, 1.0, ws, graph=graph)
        torch.cuda.synchronize()
        return 5 / (time.perf_counter() - t0)
    eager = rate(False)
    try:                                                 # same step replayed as one hipGraph per decode+gradient
        if not with_graph:
            raise RuntimeError('not measured at N > 1')
        graphed = round(rate(True), 2)
    except RuntimeError as exc:                          # capture support is the framework's, not ours: report, go on
        graphed = f'capture failed: {str(exc)[:80]}'
    return dict(value=round(eager, 2), value_hipgraph=graphed, unit='leapfrog steps/s', chains=1,
                reference_derived=3.2, note='reference: >= 2100 leapfrog decodes per image / 663 s (BASELINE.md), unstated GPU')


def settle_score_chunk(eng, make_engine, args, step, world, rank, sharding, device):
    """One untimed step on every rank before the contract's warm-up, for two reasons.  (i) The first score-network call on
    a machine fills MIOpen's on-disk kernel cache (~1 min): rank 0 goes first, alone, instead of N ranks racing through the
    same compiles and cache files.  (ii) If the step does not fit the card on ANY rank, every rank halves its score chunk
    together (a collective decision, so the ranks stay in step) and tries again."""
    import gc
    while True:
        failed = [0.0]

        def probe():
            try:
                if os.environ.get('NHMC_BENCH_FAKE_OOM') == str(rank) and not getattr(args, 'faked_oom', False):
                    args.faked_oom = True                                  # test hook: tests/test_multirank_gpu.py
                    raise torch.OutOfMemoryError('NHMC_BENCH_FAKE_OOM')
                step(eng)
                torch.cuda.synchronize()
            except torch.OutOfMemoryError:
                failed[0] = 1.0
        if world > 1:
            if rank == 0:
                probe()
            sharding.barrier()
            if rank != 0:
                probe()
        else:
            probe()
        if not sharding.max_over_ranks(failed[0], device):
            return eng
        if args.chunk <= 1:
            raise SystemExit('[bench] one chain per score call does not fit this card')
        eng = None
        gc.collect()
        torch.cuda.empty_cache()
        args.chunk = (args.chunk + 1) // 2
        if rank == 0:
            print(f'[bench] out of memory at one score chunk; every rank retries with --chunk {args.chunk}', file=sys.stderr, flush=True)
        eng = make_engine(args.chunk)


def timed_steps(eng, x, p, y, eps, sig, ws, warmup, steps, world, rank, sharding, device):
    """The contract's timed region: W untimed steps, barrier + synchronize, K steps, synchronize + barrier, max over ranks."""
    import nhmc.kernels as K

    def step():
        return eng.step(K.LF_MID, x, x, p, y, eps, sig, 1.0, ws)
    for _ in range(warmup):
        step()
    sharding.barrier()
    torch.cuda.synchronize()
    mark = torch.zeros(2, 4, device=device) if os.environ.get('NHMC_PROFILE_MARK') == '1' else None
    if mark is not None:                       # one-block marker dispatches bracketing the timed region in a rocprofv3
        K.copy_probe(mark[0], mark[1])         # kernel trace (profiles/summarize.py keeps the steady-state rows between them)
        torch.cuda.synchronize()
    eng.update_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    sharding.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, device)
    if mark is not None:
        K.copy_probe(mark[0], mark[1])
        torch.cuda.synchronize()
    events, eng.update_events = eng.update_events, None
    # per-chain loss at the end point for the final gather: a MID step does not sum it (the sampler reads the loss at a
    # trajectory's two ends only), so it is evaluated once here, outside the timed region
    _, loss, _, _ = eng.decode_and_grad(x, y)
    return dt, loss.clone(), events


def full_run_leg(device, prob, B, chunk, tau, trajectories):
    """The sampler as a user runs it: `sampler.hmc_chains` for a few COMPLETE trajectories of all B chains -- the
    once-per-run evaluation at the start point, per trajectory the schedule kernel, the momentum draw, the first half
    step from the gradient cache, L x (score ladder + data term + backward + fused update), both Hamiltonians, the
    Metropolis test, the accept commits, the cache flip and the one status read.  tau is shortened (L = floor(tau / eps))
    so the leg fits the bench's time budget; per score ladder the cost must equal the headline step's."""
    import types
    from nhmc import sampler
    x_orig = prob['x'].clamp(-1, 1)
    L = max(1, math.floor(tau / EPS))
    # With random-init weights every proposal at the configured step size is rejected (dH >> 1), and a rejected trajectory
    # skips the accept-side copies.  So half of the trajectories run at the configured (tau, eps) and half at 1/1000 of
    # both (same L, dH ~ 0: accepted), each half its own hmc_chains run: both outcomes are inside the timed region.
    legs = [(tau, EPS, trajectories - trajectories // 2), ((L + 0.5) * EPS * 1e-3, EPS * 1e-3, trajectories // 2)]
    # one untimed one-step trajectory first (the W of this leg): the first hmc_chains of a process carries a one-time
    # cost of ~6 s (measured: 23.7 s against 17.6 s for two otherwise identical runs; first-use allocations of its state,
    # sample and cache buffers next to the bench's own), not a per-trajectory one
    warm = types.SimpleNamespace(tau=1.5 * EPS, epsilon=EPS, m=1.0, sigma_0=2 * prob['sigma0'])
    sampler.hmc_chains(prob['x'], prob['b'], prob['seq'], prob['seq_next'], prob['algo'], warm, prob['y'], prob['op'], x_orig,
                       noise=sampler.PhiloxNoise(5678, 0), chunk=chunk, max_iters=1)
    dt = ladders = 0.0
    iters = accepted = rejected = 0
    per_run = []
    for tau_, eps_, n_traj in legs:
        if n_traj <= 0:
            continue
        opt = types.SimpleNamespace(tau=tau_, epsilon=eps_, m=1.0, sigma_0=2 * prob['sigma0'])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sampler.hmc_chains(prob['x'], prob['b'], prob['seq'], prob['seq_next'], prob['algo'], opt, prob['y'], prob['op'], x_orig,
                                 noise=sampler.PhiloxNoise(5678, 0), chunk=chunk, max_iters=n_traj)
        torch.cuda.synchronize()
        per_run.append(round(time.perf_counter() - t0, 2))
        dt += per_run[-1]
        assert res.L == L, (res.L, L)
        iters += res.iters
        ladders += res.ladders
        accepted += int(res.n_accept.sum())
        rejected += int(res.n_reject.sum())
    runs = sum(1 for leg in legs if leg[2] > 0)
    chunks = 1 if not chunk or chunk >= B else -(-B // chunk)
    batch_ladders = ladders / chunks                                   # batch-wide score ladders: iters * L + one per run (the start point)
    return dict(value=round(B * iters * L / dt, 3), unit='chain-steps/s', trajectories=iters, leapfrog_steps_per_trajectory=L,
                tau=tau, chains=B, seconds=round(dt, 2), runs=runs, seconds_per_run=per_run, score_ladders=batch_ladders,
                score_ladders_per_trajectory=round((batch_ladders - runs) / iters, 3),
                score_ladders_without_the_gradient_cache=iters * (L + 1),
                ms_per_ladder=round(1e3 * dt / batch_ladders, 2),
                accepted=accepted, rejected=rejected,
                note='sampler.hmc_chains end to end (prime + trajectories, every kernel and the per-trajectory status read inside '
                     'the timed region; one untimed one-step trajectory before it), half of the trajectories at the configured step size (random-init weights: all rejected) and '
                     'half at 1/1000 of it (all accepted), one run each; value = chains x trajectories x L / seconds; ms_per_ladder '
                     '(per batch-wide score ladder, the once-per-run start-point evaluations included) is comparable with ms_per_step')


def degradation_leg(device, deg, model, B, chunk, steps=2):
    """configs[2] / configs[3]: the same step with another operator -- hot path, data-term roofline, a few end-to-end steps."""
    import nhmc.kernels as K
    from nhmc import sampler, sharding
    sigma0 = {'deblur_aniso': 0.01}.get(deg, SIGMA0_CLI)               # BASELINE configs[3] quotes sigma_0 = 0.01
    prob = build_problem(device, B, 0, deg=deg, model=model, sigma0=sigma0)
    eng = sampler.LeapfrogEngine(prob['algo'].score, prob['op'], prob['b'], prob['seq'], prob['seq_next'], device, chunk=chunk)
    eps = torch.full((B,), EPS, dtype=torch.float64, device=device)
    sig = torch.full((B,), 2 * sigma0 + 1.6, dtype=torch.float64, device=device)
    ws = K.leapfrog_ws(B, CH * DIM * DIM, device)
    dt, _, _ = timed_steps(eng, prob['x'], prob['p'], prob['y'], eps, sig, ws, 1, steps, 1, 0, sharding, device)
    hot = hot_path_only(device, prob, B, 20, chunk=chunk)
    out = dict(value=round(B * steps / dt, 3), unit='chain-steps/s', steps=steps, ms_per_step=round(1e3 * dt / steps, 2),
               hot_path_only=dict(value=round(hot['value'], 1), ms_per_step=round(hot['ms_per_step'], 4)),
               roofline=data_term_roofline(device, prob, B), M=int(prob['op'].M), sigma_0=sigma0)
    if hasattr(prob['op'], 'projected') and not prob['op'].projected:
        # opt-in form (--spectral_projected): same operator, residual taken in the left singular basis
        prob['op'].projected = True
        try:
            hot = hot_path_only(device, prob, B, 20, chunk=chunk)
            out['spectral_projected'] = dict(hot_path_only=dict(value=round(hot['value'], 1), ms_per_step=round(hot['ms_per_step'], 4)),
                                             roofline=data_term_roofline(device, prob, B))
        finally:
            prob['op'].projected = False
    return out


def latent_problem(device, B, lo, chunk, seed=5678):
    """BASELINE configs[4] on one rank: LDM U-Net + VQ-f4 first stage (random init), inpaint_random at 256 x 256 on the
    decoded image, B chains of [C,64,64] latents with global chain ids lo .. lo+B-1."""
    import nhmc.kernels as K
    from nhmc import ldm, operators, plugin, sampler
    torch.manual_seed(seed)
    model = ldm.create_latent_model(ckpt=None, quiet=True).to(device)
    op = operators.build_operator('inpaint_random', CH, DIM, device, generator=torch.Generator().manual_seed(seed))
    algo = plugin.HMCLatent(model, op, 2 * SIGMA0_CLI)
    table = torch.cat([model.alphas_cumprod_prev[0:1], model.alphas_cumprod])
    seq, seq_next = [250, 500, 750], [-1, 250, 500]
    eng = sampler.LeapfrogEngine(algo.score, op, None, seq, seq_next, device, chunk=chunk, alpha_table=table,
                                 image_map=model.differentiable_decode_first_stage)
    zshape = (B, model.channels, ZDIM, ZDIM)
    x = K.randn_philox(zshape, seed, lo, 0, device=device)
    p = K.randn_philox(zshape, seed, lo, 1, device=device)
    x_true = K.randn_philox((B, CH, DIM, DIM), seed, lo, 2, device=device).clamp_(-1, 1)
    y = op.H(x_true) + (2 * SIGMA0_CLI) * torch.randn(B, op.M, device=device,
                                                       generator=torch.Generator(device=device).manual_seed(seed + lo))
    eps = torch.full((B,), 0.1, dtype=torch.float64, device=device)            # main_sampling_latent.py:828-830 default
    sig = torch.full((B,), 0.5, dtype=torch.float64, device=device)            # --sigma_y default, :832
    ws = K.leapfrog_ws(B, x[0].numel(), device)
    return dict(model=model, op=op, algo=algo, table=table, seq=seq, seq_next=seq_next, eng=eng, zshape=zshape, x=x, p=p, y=y,
                eps=eps, sig=sig, ws=ws)


def latent_leg(device, steps=6):
    """configs[4] inside the default run: a few hmc_latent steps at 16 chains (`bench.py --latent` is the full line)."""
    from nhmc import sharding
    lp = latent_problem(device, B_LATENT, 0, None)
    dt, _, _ = timed_steps(lp['eng'], lp['x'], lp['p'], lp['y'], lp['eps'], lp['sig'], lp['ws'], 2, steps, 1, 0, sharding, device)   # W = 2: the first steps allocate
    return dict(value=round(B_LATENT * steps / dt, 3), unit='chain-steps/s', steps=steps, ms_per_step=round(1e3 * dt / steps, 2),
                chains=B_LATENT, workload='BASELINE configs[4]: hmc_latent, [%d,64,64] latents, LDM U-Net + VQ-f4 decode in the loop, '
                                          'inpaint_random at 256x256, eps=0.1 sigma_y=0.5, random-init fp32' % lp['model'].channels)


def latent_main(args):
    """BASELINE configs[4]: one leapfrog step of hmc_latent = 3 x [LDM U-Net forward (no gradient: ddpm.py:892) + DDIM
    mix] + final clip + VQ codebook lookup + VQ-f4 decoder forward + inpainting data term at 256x256 + decoder backward
    + straight-through + 3 mix VJPs + fused update, 16 chains per GPU."""
    import nhmc.kernels as K
    from nhmc import ldm, operators, plugin, sampler, sharding
    rank, local_rank, world = sharding.init_process_group('gloo' if args.rehearse_shared_gpu else None)
    local_rank = 0 if args.rehearse_shared_gpu else local_rank
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    B = args.batch or B_LATENT
    lp = latent_problem(device, B, rank * B, args.chunk)
    model, op, algo, table, seq, seq_next, eng = (lp[k] for k in ('model', 'op', 'algo', 'table', 'seq', 'seq_next', 'eng'))
    x, p, y, eps, sig, ws, zshape = (lp[k] for k in ('x', 'p', 'y', 'eps', 'sig', 'ws', 'zshape'))
    args.chunk = args.chunk or B
    eng = settle_score_chunk(eng, lambda c: sampler.LeapfrogEngine(algo.score, op, None, seq, seq_next, device, chunk=c, alpha_table=table,
                                                                  image_map=model.differentiable_decode_first_stage),
                             args, lambda e: e.step(K.LF_MID, x, x, p, y, eps, sig, 1.0, ws), world, rank, sharding, device)
    dt, loss, _ = timed_steps(eng, x, p, y, eps, sig, ws, args.warmup, args.steps, world, rank, sharding, device)
    stats = torch.stack([loss.float(), x.reshape(B, -1).pow(2).sum(1)], dim=1).contiguous()
    allstats = sharding.gather_chains(stats, world * B, rank, world)
    if rank == 0:
        roof = leapfrog_roofline(device, B_PER_GPU, args.roofline_launches)
        # the HIP kernel this path adds: codebook lookup of all 16 x 4096 latent pixels against 8192 codes
        z = K.randn_philox(zshape, 3, 0, 0, device=device).clamp_(-1, 1)
        cb = model.first_stage_model.quantize.embedding.weight.detach().contiguous()
        for _ in range(3):
            K.vq_nearest(z, cb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            K.vq_nearest(z, cb)
        e1.record()
        torch.cuda.synchronize()
        vq_us = e0.elapsed_time(e1) * 1e3 / 50
        n_pairs = B * ZDIM * ZDIM * cb.shape[0]
        line = {
            'metric': 'HMC leapfrog chain-steps/sec (FFHQ-latent 64x64x3 hmc_latent, inpaint_random at 256x256, LDM U-Net + VQ-f4 decode in the loop)',
            'value': round(world * B * args.steps / dt, 3), 'unit': 'chain-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 2), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[4]: FFHQ-latent [{model.channels},64,64] (the reference config has 3 latent '
                                   f'channels; BASELINE.json says 4), hmc_latent, inpaint_random sigma_0=0.05 eps=0.1 sigma_y=0.5 '
                                   f'timesteps=3, {B} chains per GPU, LDM U-Net (224 ch) + VQ-f4 decoder (128 ch, 8192 codes) random-init fp32',
                       'chains_per_gpu': B, 'global_chains': world * B, 'score_chunk': args.chunk or B,
                       'parallelism': f'chains sharded over {world} rank(s), no data-path collective'},
            'roofline': dict(bound='hbm', achieved=round(roof['achieved'], 1), peak=HBM_PEAK_GBS, unit='GB/s',
                             frac=round(roof['achieved'] / HBM_PEAK_GBS, 4), traffic=None, kernel=roof['kernel'],
                             avg_us=round(roof['avg_us'], 2),
                             note='the fused update at the pixel path\'s 64 x 196 608 elements (the latent state is 16 x 12 288: launch-bound)'),
            'vq_kernel': dict(kernel='k_vq_nearest<3> (nhmc_vq_nearest)', avg_us=round(vq_us, 1), pairs=n_pairs,
                              gpairs_per_s=round(n_pairs / vq_us / 1e3, 1),
                              note='one thread per latent pixel, codebook staged through LDS; torch would materialise a '
                                   f'{n_pairs * 4 / 2 ** 30:.1f} GiB distance matrix'),
            'final_gather': dict(chains=int(allstats.shape[0]), loss_mean=float(allstats[:, 0].double().mean())),
            'cpu_baseline': None,
        }
        print(json.dumps(line), flush=True)
    sharding.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` on its own: start the N ranks as a CHILD torch.distributed.run (one process per GPU)
    with the same arguments and hand back its exit code.  Called before anything in this process touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] --gpus %d without WORLD_SIZE: launching %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd).returncode


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs a GPU; the HIP path has no CPU fallback'
    if args.latent:
        return latent_main(args)
    import nhmc.kernels as K
    from nhmc import sampler, sharding
    if args.rehearse_shared_gpu:
        rank, local_rank, world = sharding.init_process_group('gloo')
        local_rank = 0
    else:
        rank, local_rank, world = sharding.init_process_group()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    B = args.batch or B_PER_GPU
    lo = rank * B                                                      # global chain ids of this rank (weak scaling)
    prob = build_problem(device, B, lo, deg=args.deg, tiny=args.tiny_score)
    if args.chunk is None:
        # score chunk = the whole batch when its three autograd graphs fit: measured 3.45 GiB per chain with the fused
        # GroupNorm kernels (220.8 GiB peak at 64 chains); otherwise halves.  Same decision on every rank (same cards).
        free = torch.cuda.mem_get_info(device)[0] / 2 ** 30
        args.chunk = B
        while args.chunk > 1 and not args.tiny_score and 3.45 * 1.12 * args.chunk > free:
            args.chunk = (args.chunk + 1) // 2
    eng = sampler.LeapfrogEngine(prob['algo'].score, prob['op'], prob['b'], prob['seq'], prob['seq_next'], device,
                                 chunk=args.chunk)
    x, p, y = prob['x'], prob['p'], prob['y']
    N = x[0].numel()
    eps = torch.full((B,), EPS, dtype=torch.float64, device=device)
    sig = torch.full((B,), 2 * SIGMA0_CLI + 1.6, dtype=torch.float64, device=device)     # sigma_y at epoch 0
    ws = K.leapfrog_ws(B, N, device)

    ms_per_step = value = gather = None
    in_situ = []
    if not args.kernel_only:
        eng = settle_score_chunk(eng, lambda c: sampler.LeapfrogEngine(prob['algo'].score, prob['op'], prob['b'], prob['seq'],
                                                                      prob['seq_next'], device, chunk=c),
                                 args, lambda e: e.step(K.LF_MID, x, x, p, y, eps, sig, 1.0, ws), world, rank, sharding, device)
        dt, loss, in_situ = timed_steps(eng, x, p, y, eps, sig, ws, args.warmup, args.steps, world, rank, sharding, device)
        ms_per_step = 1e3 * dt / args.steps
        value = world * B * args.steps / dt
        # the one collective of the design: per-chain results gathered once, after the timed region (RCCL over xGMI at N > 1)
        stats = torch.stack([loss.float(), x.reshape(B, -1).pow(2).sum(1)], dim=1).contiguous()
        torch.cuda.synchronize()
        sharding.barrier()
        t0 = time.perf_counter()                                   # times the collective alone
        allstats = sharding.gather_chains(stats, world * B, rank, world)
        torch.cuda.synchronize()
        gather = dict(chains=int(allstats.shape[0]), ms=round(1e3 * (time.perf_counter() - t0), 3),
                      loss_mean=float(allstats[:, 0].double().mean()))
    roof = hot = cpu = single = by_deg = full = None
    if rank == 0:
        roof = leapfrog_roofline(device, B, args.roofline_launches)
        hot = hot_path_only(device, prob, B, 20, chunk=args.chunk)
        ktable = kernel_table(device, prob, B) if args.deg == 'inpaint_random' else None
        if in_situ:
            us = [a.elapsed_time(b_) * 1e3 for a, b_, _ in in_situ]
            chains = sum(n for _, _, n in in_situ) / len(in_situ)
            roof['in_situ_us'] = sum(us) / len(us)
            roof['in_situ_chains_per_launch'] = chains
            roof['in_situ_gbs'] = 6 * chains * N * 4 / (roof['in_situ_us'] * 1e-6) / 1e9
            roof['in_situ_frac'] = roof['in_situ_gbs'] / HBM_PEAK_GBS
            roof['in_situ_note'] = ('per-launch event pairs inside the timed steps (one launch per score chunk), with the second '
                                    'gradient pointer (R x,p,g,g2 + W x,p = 6T = 24 B/element), caches cold after the score network')
        traffic = source = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_leapfrog.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                rec = json.load(f)
            traffic, source = rec.get('hbm_bytes_per_launch'), rec.get('source', 'profiles/traffic_leapfrog.json')
        roofline = dict(bound='hbm', achieved=round(roof['achieved'], 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(roof['achieved'] / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=source,
                        **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in roof.items() if k != 'achieved'})
        if world == 1 and not args.kernel_only and not args.no_by_deg and args.deg == 'inpaint_random':
            by_deg = {}
            for deg in ('sr4', 'deblur_aniso'):
                by_deg[deg] = degradation_leg(device, deg, prob['model'], B, args.chunk)
        if world == 1 and not args.kernel_only and not args.no_full_run:
            full = full_run_leg(device, prob, B, args.chunk, args.full_run_tau, args.full_run_trajectories)
            if ms_per_step:
                full['ms_per_ladder_over_ms_per_step'] = round(full['ms_per_ladder'] / ms_per_step, 4)
        if not args.kernel_only:
            single = single_chain_rate(eng, x, p, y, eps, sig, with_graph=(world == 1))
        if by_deg is not None and not args.tiny_score:
            prob['algo'] = prob['model'] = eng = None                       # the pixel-space network is done
            torch.cuda.empty_cache()
            by_deg['hmc_latent'] = latent_leg(device)
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        line = {
            'metric': f'HMC leapfrog chain-steps/sec (256x256x3 FFHQ {args.deg}, U-Net score in the loop)',
            'value': None if value is None else round(value, 3), 'unit': 'chain-steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': None if ms_per_step is None else round(ms_per_step, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('BASELINE configs[1]' if args.deg == 'inpaint_random' else 'BASELINE configs[1] with another degradation') +
                                   f': FFHQ 256x256 {args.deg} sigma_0=0.05 tau=1.0 eps=0.05 '
                                   f'timesteps=3, {B} chains per GPU, ' +
                                   ('REHEARSAL with a 32-channel U-Net (not the metric)' if args.tiny_score else 'FFHQ U-Net architecture random-init fp32') +
                                   (' -- REHEARSAL: the ranks share ONE GPU over gloo (not a multi-GPU measurement)' if args.rehearse_shared_gpu else ''),
                       'chains_per_gpu': B, 'global_chains': world * B, 'score_chunk': args.chunk,
                       'parallelism': f'chains sharded over {world} rank(s), no data-path collective'},
            'roofline': roofline, 'hot_path_only': hot, 'hot_path_kernels': ktable, 'full_run': full, 'by_deg': by_deg, 'single_chain': single, 'final_gather': gather,
            'peak_memory_gib': round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1), 'cpu_baseline': cpu,
        }
        print(json.dumps(line), flush=True)
    sharding.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
