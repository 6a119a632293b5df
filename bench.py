#!/usr/bin/env python3
"""bench.py -- HMC leapfrog chain-steps/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): FFHQ 256x256x3, deg = inpaint_random (M = 15 729), sigma_0 = 0.05,
tau = 1.0, eps = 0.05, timesteps = 3, 64 chains PER GPU (weak scaling: chains are sharded, never data),
synthetic inputs, guided-diffusion FFHQ U-Net architecture with random-init fp32 weights (the checkpoint
is fetch-only; random init is also the reference's own fallback).

A "step" is one leapfrog step of every chain of the batch: the decode (3 x [score forward + DDIM mix]),
the data term, the backward (3 x [DDIM mix VJP + score input-gradient]) and the fused momentum+position
update -- nothing is skipped or cached.  value = N * 64 * K / (max-over-ranks time).

Extra objects on the same line:
  roofline      the dominant HIP kernel (fused leapfrog update, 20 B/element = 5T per chain):
                algorithmic bytes per launch / average launch duration, HIP events on the launch stream
                over a region of back-to-back launches that rotates over buffer sets larger than the
                256 MiB Infinity Cache, so every launch streams from HBM as it does between two score
                evaluations.  `in_situ_us` is the same kernel timed per launch inside the timed steps.
  hot_path      the HIP-side of one step alone (score replaced by a resident tensor): chain-steps/s
  cpu_baseline  the oracle (CPU restatement, validated bit-exact against the reference) doing the same
                step on the host cores: B = 1, one step, same U-Net architecture.
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
B_PER_GPU, DIM, CH = 64, 256, 3
SIGMA0_CLI, TAU, EPS, TIMESTEPS = 0.05, 1.0, 0.05, 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--chunk', type=int, default=32, help='chains per score-network call (activation memory)')
    ap.add_argument('--batch', type=int, default=B_PER_GPU, help='chains per GPU (BASELINE: 64)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--kernel-only', action='store_true', help='skip the end-to-end steps (profiling the HIP kernels)')
    ap.add_argument('--roofline-launches', type=int, default=200)
    ap.add_argument('--deg', default='inpaint_random',
                    help='degradation: inpaint_random (the BASELINE metric), sr4 (configs[2]), deblur_aniso (configs[3]), ...')
    ap.add_argument('--rehearse-shared-gpu', action='store_true',
                    help='rehearsal only: all ranks use cuda:0 over gloo (checks the N>1 control flow on a 1-GPU box)')
    return ap.parse_args()


def build_problem(device, B, chain_id0, seed=5678, deg='inpaint_random'):
    import nhmc.kernels as K
    from nhmc import operators, plugin, sampler, schedule, unet
    gen = torch.Generator().manual_seed(seed)
    op = operators.build_operator(deg, CH, DIM, device, generator=gen)
    torch.manual_seed(seed)
    model = unet.create_model(**unet.FFHQ_CONFIG).to(device).eval().requires_grad_(False)
    algo = plugin.HMC(model, op, 2 * SIGMA0_CLI)
    b = torch.from_numpy(schedule.get_beta_schedule('linear', beta_start=1e-4, beta_end=0.02,
                                                    num_diffusion_timesteps=1000)).float().to(device)
    seq, seq_next = schedule.timestep_ladder(1000, TIMESTEPS)
    shape = (B, CH, DIM, DIM)
    x = K.randn_philox(shape, seed, chain_id0, 0, device=device)
    p = K.randn_philox(shape, seed, chain_id0, 1, device=device)
    x_true = K.randn_philox(shape, seed, chain_id0, 2, device=device).clamp_(-1, 1)
    y = op.H(x_true) + (2 * SIGMA0_CLI) * torch.randn(B, op.M, device=device,
                                                       generator=torch.Generator(device=device).manual_seed(seed + chain_id0))
    return dict(op=op, algo=algo, b=b, seq=seq, seq_next=seq_next, x=x, p=p, y=y, model=model)


def leapfrog_roofline(device, B, launches):
    """Dominant kernel: nhmc_leapfrog_fused(MID).  Rotates over R buffer sets (x,p,g) whose total
    footprint exceeds the Infinity Cache, events around the whole region on the launch stream."""
    import nhmc.kernels as K
    N = CH * DIM * DIM
    per_set = 3 * B * N * 4
    R = max(2, math.ceil(3 * (256 << 20) / per_set))                 # >= 3x the 256 MiB Infinity Cache
    sets = [tuple(K.randn_philox((B, CH, DIM, DIM), 1, 0, 3 * r + k, device=device) for k in range(3)) for r in range(R)]
    eps = torch.full((B,), 1e-3, dtype=torch.float64, device=device)
    sig = torch.full((B,), 1.7, dtype=torch.float64, device=device)
    for r in range(R):                                               # warm-up (code object load, TLB)
        K.leapfrog_fused(K.LF_MID, sets[r][0], sets[r][1], sets[r][2], eps, sig, 1.0)
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
                achieved=alg_bytes / avg_s / 1e9, launches=launches, buffer_sets=R,
                footprint_mib=R * per_set / 2 ** 20, copy_kernel_gbs=copy_gbs,
                frac_of_copy_kernel=alg_bytes / avg_s / 1e9 / copy_gbs,
                copy_note='nhmc_copy_probe: 1 read : 1 write streaming copy, same buffers; the update is 3 reads : 2 writes')


def hot_path_only(device, prob, B, steps):
    """The HIP side of one leapfrog step with the score output held resident (kernel-comparable number)."""
    import nhmc.kernels as K
    op = prob['op']
    x, p, y = prob['x'].clone(), prob['p'].clone(), prob['y']
    e = K.randn_philox((B, 2 * CH, DIM, DIM), 7, 0, 0, device=device)
    gs = K.randn_philox((B, CH, DIM, DIM), 7, 0, 1, scale=1e-3, device=device)
    ge = [torch.zeros_like(e) for _ in range(3)]            # persistent score-gradient buffers (sigma-channels stay zero)
    at =[torch.tensor([a], device=device).expand(B).contiguous() for a in (0.0033001585, 0.0777966604, 0.5214230418)]
    atn = [torch.tensor([a], device=device).expand(B).contiguous() for a in (0.0777966604, 0.5214230418, 1.0)]
    eps = torch.full((B,), EPS, dtype=torch.float64, device=device)
    sig = torch.full((B,), 1.7, dtype=torch.float64, device=device)

    def step():
        cur, ins = x, []
        for s in range(3):
            ins.append(cur)
            cur = K.ddim_mix_fwd(cur, e, at[s], atn[s], final_clip=(s == 2))['xt_next']
        g2 = None
        for s in (2, 1, 0):
            if s == 2 and hasattr(op, 'fused_last_vjp'):            # data term fused into the last-step VJP
                extra = dict(xt_next=cur) if getattr(op, 'fused_wants_decode', False) else {}
                loss, g, g_e = op.fused_last_vjp(ins[s], e, at[s], atn[s], y, g_e_out=ge[s], **extra)
            elif s == 2:
                loss, g = op.data_term(cur, y, apply_clip=False)
                g, g_e = K.ddim_mix_bwd(g, ins[s], e, at[s], atn[s], final_clip=True, g_e_out=ge[s])
            else:
                g, g_e = K.ddim_mix_bwd(g, ins[s], e, at[s], atn[s], gout2=g2, g_e_out=ge[s])
            g2 = gs                                                 # stands in for the score's input-gradient
        K.leapfrog_fused(K.LF_MID, x, p, g, eps, sig, 1.0, g2=g2)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(value=B * steps / dt, unit='chain-steps/s', ms_per_step=1e3 * dt / steps, steps=steps,
                note='score output resident (no U-Net); 3 mix fwd + (data term + last VJP fused) + 2 mix VJP + fused update')


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
    """Oracle leapfrog step on the host cores: B = 1, same architecture, one step (~10-30 s)."""
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
    x = torch.randn(1, CH, DIM, DIM, generator=gen)
    p = torch.randn(1, CH, DIM, DIM, generator=gen)
    y = op.H(torch.rand(1, CH, DIM, DIM, generator=gen) * 2 - 1) + 0.1 * torch.randn(1, op.M, generator=gen)
    steps = 4                                                                       # ~15-20 s on the box's 16-CPU quota
    t0 = time.perf_counter()
    for _ in range(steps):
        xl = x.clone().requires_grad_(True)
        _, _, _, g = hmc_ref._data_loss_and_grad(xl, b, seq, seq_next, net, op, y)  # decode + gradient
        x, p = hmc_ref.leapfrog_update('mid', x, p, g, eps=EPS, sigma_y=1.7, m=1.0)[:2]   # momentum + position
    dt = time.perf_counter() - t0
    return dict(value=steps / dt, unit='chain-steps/s', cores=torch.get_num_threads(), kind='port',
                sample=f'oracle (oracle/hmc_ref.py) leapfrog steps, B=1, {steps} consecutive steps, FFHQ U-Net fp32 on CPU: {dt:.1f} s')


def single_chain_rate(eng, x, p, y, eps, sig, with_graph):
    """The reference's own operating point: ONE chain (its hmc() is batch-1 only); BASELINE.md derives ~3.2 leapfrog
    steps/s for it from the authors' logs (unstated NVIDIA GPU).  Runs last: a failed graph capture must not disturb
    the measurements above."""
    import nhmc.kernels as K
    x1, p1, y1 = x[:1].clone(), p[:1].clone(), y[:1].contiguous()
    e1, s1 = eps[:1].contiguous(), sig[:1].contiguous()
    for _ in range(2):
        _, _, ga, gb = eng.decode_and_grad(x1, y1)
        K.leapfrog_fused(K.LF_MID, x1, p1, ga, e1, s1, 1.0, g2=gb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        _, _, ga, gb = eng.decode_and_grad(x1, y1)
        K.leapfrog_fused(K.LF_MID, x1, p1, ga, e1, s1, 1.0, g2=gb)
    torch.cuda.synchronize()
    eager = 5 / (time.perf_counter() - t0)
    graphed = None
    try:                                                 # same step replayed as one hipGraph per decode+gradient
        if not with_graph:
            raise RuntimeError('not measured at N > 1')
        for _ in range(2):
            _, _, ga, gb = eng.decode_and_grad(x1, y1, graph=True)
            K.leapfrog_fused(K.LF_MID, x1, p1, ga, e1, s1, 1.0, g2=gb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            _, _, ga, gb = eng.decode_and_grad(x1, y1, graph=True)
            K.leapfrog_fused(K.LF_MID, x1, p1, ga, e1, s1, 1.0, g2=gb)
        torch.cuda.synchronize()
        graphed = round(5 / (time.perf_counter() - t0), 2)
    except RuntimeError as exc:                          # capture support is the framework's, not ours: report, go on
        graphed = f'capture failed: {str(exc)[:80]}'
    single = dict(value=round(eager, 2), value_hipgraph=graphed, unit='leapfrog steps/s', chains=1,
                  reference_derived=3.2, note='reference: >= 2100 leapfrog decodes per image / 663 s (BASELINE.md), unstated GPU')
    return single


def main():
    args = parse()
    import nhmc.kernels as K
    from nhmc import sampler, sharding
    assert torch.cuda.is_available(), 'bench.py needs a GPU; the HIP path has no CPU fallback'
    if args.rehearse_shared_gpu:
        rank, local_rank, world = sharding.init_process_group('gloo')
        local_rank = 0
    else:
        rank, local_rank, world = sharding.init_process_group()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    B = args.batch
    lo = rank * B                                                      # global chain ids of this rank (weak scaling)
    prob = build_problem(device, B, lo, deg=args.deg)
    eng = sampler.LeapfrogEngine(prob['algo'].score, prob['op'], prob['b'], prob['seq'], prob['seq_next'], device,
                                 chunk=args.chunk)
    x, p, y = prob['x'], prob['p'], prob['y']
    eps = torch.full((B,), EPS, dtype=torch.float64, device=device)
    sig = torch.full((B,), 2 * SIGMA0_CLI + 1.6, dtype=torch.float64, device=device)     # sigma_y at epoch 0
    in_situ = []
    last = {}

    def step(timed):
        xt, loss, ga, gb = eng.decode_and_grad(x, y)
        last['loss'] = loss
        if timed:
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        K.leapfrog_fused(K.LF_MID, x, p, ga, eps, sig, 1.0, g2=gb)
        if timed:
            b_.record()
            in_situ.append((a, b_))

    ms_per_step = value = None
    if not args.kernel_only:
        if world > 1:
            # the first score-network call on a machine fills MIOpen's on-disk kernel cache (~1 min); let one rank do it
            # instead of N ranks racing through the same compiles and the same cache files
            if rank == 0:
                step(False)
                torch.cuda.synchronize()
            sharding.barrier()
        for _ in range(args.warmup):
            step(False)
        sharding.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        torch.cuda.synchronize()
        sharding.barrier()
        dt = sharding.max_over_ranks(time.perf_counter() - t0, device)
        ms_per_step = 1e3 * dt / args.steps
        value = world * B * args.steps / dt

    # the one collective of the design: per-chain results gathered once, after the timed region (RCCL over xGMI at N > 1)
    gather = None
    if not args.kernel_only:
        stats = torch.stack([last['loss'].float(), x.reshape(B, -1).pow(2).sum(1)], dim=1).contiguous()
        torch.cuda.synchronize()
        sharding.barrier()
        t0 = time.perf_counter()                                   # times the collective alone
        if args.rehearse_shared_gpu:
            allstats = sharding.gather_chains(stats.cpu(), world * B, rank, world)
        else:
            allstats = sharding.gather_chains(stats, world * B, rank, world)
        torch.cuda.synchronize()
        gather = dict(chains=int(allstats.shape[0]), ms=round(1e3 * (time.perf_counter() - t0), 3),
                      loss_mean=float(allstats[:, 0].double().mean()))
    roof = hot = cpu = single = None
    if rank == 0:
        roof = leapfrog_roofline(device, B, args.roofline_launches)
        hot = hot_path_only(device, prob, B, 20)
        if in_situ:
            roof['in_situ_us'] = sum(a.elapsed_time(b_) for a, b_ in in_situ) * 1e3 / len(in_situ)
            roof['in_situ_gbs'] = 6 * B * CH * DIM * DIM * 4 / (roof['in_situ_us'] * 1e-6) / 1e9
            roof['in_situ_frac'] = roof['in_situ_gbs'] / HBM_PEAK_GBS
            roof['in_situ_note'] = ('per-launch event pairs inside the timed steps; there the kernel takes the second gradient '
                                    'pointer (R x,p,g,g2 + W x,p = 6T = 24 B/element), caches cold after the score network')
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_leapfrog.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get('hbm_bytes_per_launch')
        roofline = dict(bound='hbm', achieved=round(roof['achieved'], 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(roof['achieved'] / HBM_PEAK_GBS, 4), traffic=traffic,
                        **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in roof.items() if k != 'achieved'})
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        if not args.kernel_only:
            single = single_chain_rate(eng, x, p, y, eps, sig, with_graph=(world == 1))
        line = {
            'metric': f'HMC leapfrog chain-steps/sec (256x256x3 FFHQ {args.deg}, U-Net score in the loop)',
            'value': None if value is None else round(value, 3), 'unit': 'chain-steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': None if ms_per_step is None else round(ms_per_step, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('BASELINE configs[1]' if args.deg == 'inpaint_random' else 'BASELINE configs[1] with another degradation') +
                                   f': FFHQ 256x256 {args.deg} sigma_0=0.05 tau=1.0 eps=0.05 '
                                   f'timesteps=3, {B} chains per GPU, FFHQ U-Net architecture random-init fp32',
                       'chains_per_gpu': B, 'global_chains': world * B, 'score_chunk': args.chunk,
                       'parallelism': f'chains sharded over {world} rank(s), no data-path collective'},
            'roofline': roofline, 'hot_path_only': hot, 'single_chain': single, 'final_gather': gather, 'cpu_baseline': cpu,
        }
        print(json.dumps(line), flush=True)
    sharding.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
