"""Command line with the reference's flags (`get_parser`, main_sampling.py:923-1015) for the HMC path.

    python main_sampling.py --dataset ffhq --algo hmc --timesteps 3 --deg inpaint_random --sigma_0 0.05 \
        -i exp/samples/ffhq/inpaint_random/hmc --tau 1.0 --epsilon 0.05          (README.md:79 of the reference)

Same flag names and defaults; unknown flags are ignored as in the reference (`parse_known_args`, :1031).  Only
`--algo hmc` (and `--algo hmc_latent` through `main_latent`, the counterpart of main_sampling_latent.py:791-918 with
its own defaults: `--epsilon 0.1`, no `--annealed_temp`) and the degradations on the hot path are served; anything
else raises `NotImplementedError` (the reference's own error for an unknown algo, :256-257).  Additions, all optional: `--chains` images are
sampled in parallel as independent chains (the reference is batch-1), `--score_chunk` bounds the U-Net batch,
`--synthetic K` uses K synthetic images when no dataset folder is present, `--philox` switches the noise to the
shard-invariant counter-based generator.  Under torchrun the images are sharded over ranks and the per-image
metrics are gathered once at the end (RCCL).  The measurement noise and the start point of image s are drawn from a
generator keyed by (seed, s) -- never from the global stream -- so with `--philox` (implied when WORLD_SIZE > 1) an
image's result does not depend on the number of ranks (nor, up to the score network's batch-size-dependent
convolution rounding, on `--chains`).  `--spectral_projected` switches the two blur operators to the four-product data
term (operators.Deblurring2D).
"""
import argparse
import glob
import os
import sys
import random

import numpy as np
import torch

from . import ldm, operators, plugin, sampler, schedule, sharding, unet
from . import kernels as K


def get_parser(latent=False):
    """`get_parser` of main_sampling.py:923-1015, or of main_sampling_latent.py:791-897 with latent=True."""
    p = argparse.ArgumentParser()
    p.add_argument('--seed', type=int, default=5678, help='Random seed')
    p.add_argument('--exp', type=str, default='exp', help='Path for saving running related data.')
    p.add_argument('--dataset', type=str, nargs='?', default='celeba', help='dataset')
    p.add_argument('--default_lr', action='store_true')
    p.add_argument('--lr', type=float, nargs='?', default=1.0, help='Step-Size')
    p.add_argument('--N', type=int, default=1, help='N repeats')
    p.add_argument('--deg', type=str, required=True, help='Degradation')
    p.add_argument('--sigma_0', type=float, required=True, help='Sigma_0')
    p.add_argument('--tau', type=float, default=1.0, help='Tau for HMC')
    p.add_argument('--epsilon', type=float, default=0.1 if latent else 0.05, help='Epsilon for HMC')
    p.add_argument('--sigma_y', type=float, default=0.5, help='sigma_y for HMC (measurement noise)')
    p.add_argument('--m', type=float, default=1.0, help='Mass Matrix Variance')
    if not latent:
        p.add_argument('--annealed_temp', action='store_true', default=False)
    p.add_argument('--noise', type=str, default='ddpm', help='Type of Noise')
    p.add_argument('--num_timesteps', type=int, nargs='?', default=1000)
    p.add_argument('--timesteps', type=int, nargs='?', default=10)
    p.add_argument('--subset_start', type=int, default=-1)
    p.add_argument('--subset_end', type=int, default=-1)
    p.add_argument('--algo', type=str, nargs='?', default='resample')
    p.add_argument('--refine', action='store_true')
    p.add_argument('-i', '--image_folder', type=str, default='exp/samples/ffhq/00000')
    # additions
    p.add_argument('--chains', type=int, default=1, help='images sampled in parallel (independent chains)')
    p.add_argument('--score_chunk', type=int, default=None,
                   help='chains per score-network call; default: as many of --chains as the free device memory holds')
    p.add_argument('--synthetic', type=int, default=0, help='use this many synthetic images')
    p.add_argument('--philox', action='store_true', help='counter-based, shard-invariant noise')
    p.add_argument('--save_images', action='store_true')
    p.add_argument('--spectral_projected', action='store_true',
                   help='deblur_aniso / deblur_gauss: residual in the left singular basis, 4 products instead of 8 (rounds differently)')
    p.add_argument('--graph', dest='use_graph', action='store_true',
                   help='replay each decode+gradient chunk as a hipGraph (helps single-chain runs: ~1.2x)')
    p.add_argument('--hmc_epochs', type=int, default=60, help='annealing epochs (reference: 60, main_sampling.py:665)')
    p.add_argument('--hmc_sampling', type=int, default=20, help='collected samples (reference: 20, :666)')
    return p


FFHQ_DEFAULTS = {
    'data': {'dataset': 'ffhq', 'image_size': 256, 'channels': 3, 'rescaled': True},
    'model': dict(unet.FFHQ_CONFIG, model_path='models/ffhq_10m.pt'),
    'diffusion': {'beta_schedule': 'linear', 'beta_start': 1e-4, 'beta_end': 0.02, 'num_diffusion_timesteps': 1000},
}


FFHQ_LATENT_DEFAULTS = {
    'data': {'dataset': 'ffhq', 'image_size': 256, 'channels': 3, 'rescaled': True},
    'model': {'target': 'ldm.models.diffusion.ddpm.LatentDiffusion', 'params': ldm.FFHQ_LDM},
}


def load_config(dataset, latent=False):
    """configs/config_{dataset}.yml (main_sampling.py:1033) or configs/config_{dataset}_latent.yml
    (main_sampling_latent.py:904); the FFHQ values are built in for a checkout without the reference's configs/."""
    path = f'configs/config_{dataset}_latent.yml' if latent else f'configs/config_{dataset}.yml'
    if os.path.exists(path):
        import yaml
        with open(path) as f:
            return yaml.safe_load(f)
    if dataset == 'ffhq':
        return FFHQ_LATENT_DEFAULTS if latent else FFHQ_DEFAULTS
    raise FileNotFoundError(path)


def load_images(folder, size, start, end, synthetic, seed):
    files = sorted(glob.glob(os.path.join(folder, '**', '*.png'), recursive=True)) if os.path.isdir(folder) else []
    if files and not synthetic:
        from PIL import Image
        if start >= 0 and end > 0:
            files = files[start:end]
        imgs = [torch.from_numpy(np.asarray(Image.open(f).convert('RGB').resize((size, size)), dtype=np.float32) / 255.0)
                .permute(2, 0, 1) for f in files]
        return torch.stack(imgs) * 2 - 1                                     # data_transform: rescaled -> [-1,1]
    n = synthetic or 1
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(n, 3, size // 16, size // 16, generator=g)
    return torch.nn.functional.interpolate(low, size=size, mode='bicubic', align_corners=False).clamp(0, 1) * 2 - 1


def auto_score_chunk(opt, device, probe, world=1):
    """Chains per score call when --score_chunk is not given.  Called AFTER the networks are on the device: `probe()` runs
    one decode + gradient of ONE chain through the engine the run will use (pixel path: three U-Net autograd graphs,
    3.45 GiB at FFHQ size; latent path: a no-grad LDM U-Net ladder + the VQ-f4 decoder's graph at 256 x 256), the peak it
    adds is the per-chain cost, and the free memory left -- divided by the ranks that share the card in a shared-GPU
    rehearsal (NHMC_SHARED_GPU=1) -- is filled to 1 / 1.12."""
    if opt.score_chunk:
        return opt.score_chunk
    torch.cuda.synchronize(device)
    torch.cuda.reset_peak_memory_stats(device)
    base = torch.cuda.memory_allocated(device)
    probe()
    torch.cuda.synchronize(device)
    per_chain = max(1, torch.cuda.max_memory_allocated(device) - base)
    torch.cuda.empty_cache()
    sharers = world if os.environ.get('NHMC_SHARED_GPU') == '1' else 1
    free = torch.cuda.mem_get_info(device)[0] / sharers
    return max(1, min(opt.chains, int(free / (per_chain * 1.12))))


def run_with_oom_backoff(opt, run):
    """run() with the score chunk halved after an out-of-memory error (a wrong estimate must not abort a run that
    takes hours).  Local to the rank: the sampler has no collective inside a run, so ranks need not agree."""
    import gc
    while True:
        try:
            return run()
        except torch.OutOfMemoryError:
            chunk = opt.score_chunk or opt.chains
            if chunk <= 1:
                raise
            opt.score_chunk = (chunk + 1) // 2
            gc.collect()
            torch.cuda.empty_cache()
            print(f'[nhmc] out of memory at {chunk} chains per score call; retrying with --score_chunk {opt.score_chunk}',
                  file=sys.stderr, flush=True)


def image_generator(seed, s):
    """Generator of image s: measurement noise and start point are keyed by (seed, global image index), on the host,
    so they do not depend on the rank that samples the image, on WORLD_SIZE or on --chains."""
    return torch.Generator().manual_seed((int(seed) * 1000003 + int(s)) & 0x7FFFFFFFFFFFFFFF)


def draw_inputs(seed, s, y_clean, sigma_0, x_shape):
    """-> (y_0, x_start) for image s: y_0 = H(x_orig) + sigma_0 * N(0,1) (main_sampling.py:447-448) and the
    N(0,1) start point (:460-466), both from `image_generator(seed, s)`.  y_clean: [M] on any device."""
    g = image_generator(seed, s)
    noise = torch.randn(y_clean.shape, generator=g)
    x = torch.randn(x_shape, generator=g)
    return y_clean + sigma_0 * noise.to(y_clean.device), x.to(y_clean.device)


def image_batches(n_images, rank, world, chains):
    """Global image indices this rank samples, in batches of `chains`: contiguous block partition over ranks
    (sharding.chain_range), batches never straddle ranks."""
    lo, hi = sharding.chain_range(n_images, rank, world)
    return [list(range(s, min(hi, s + chains))) for s in range(lo, hi, chains)]


def _setup(opt, latent):
    config = load_config(opt.dataset, latent)
    rank, local_rank, world = sharding.init_process_group()
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(opt.seed)
    np.random.seed(opt.seed)
    random.seed(opt.seed)
    size, ch = config['data']['image_size'], config['data']['channels']
    # mask = first torch RNG draw, as in the reference
    op = operators.build_operator(opt.deg, ch, size, device, spectral_projected=opt.spectral_projected or None)
    opt.sigma_0 = 2 * opt.sigma_0                                           # [-1,1] scaling, main_sampling.py:348
    if world > 1 and not opt.philox:
        if rank == 0:
            print('WORLD_SIZE > 1: switching to the counter-based noise (--philox) so results do not depend on the sharding')
        opt.philox = True
    if opt.philox:
        opt.philox_seed = opt.seed
    opt.quiet = opt.chains > 1 or rank != 0
    opt.progress_every = 10 if rank == 0 else 0                           # stderr heartbeat for long quiet runs
    skip = opt.num_timesteps // (opt.timesteps + 1)                          # main_sampling.py:469-471
    seq = list(range(skip, opt.num_timesteps, skip))
    images = load_images(os.path.join(opt.exp, 'datasets', opt.dataset), size, opt.subset_start, opt.subset_end,
                         opt.synthetic, opt.seed)
    return config, rank, world, device, op, seq, [-1] + seq[:-1], images


def _report(rows, n_images, rank, world, device):
    local = torch.tensor(rows, dtype=torch.float32, device=device).reshape(-1, 3)
    table = sharding.gather_chains(local, n_images, rank, world).cpu()
    if rank == 0:
        for idx, mean, std in table.tolist():
            print(f'image {int(idx)}: PSNR {mean:.3f} (std over samples {std:.4f})' if mean == mean else
                  f'image {int(idx)}: no sample was collected (every proposal of the final phase was rejected)')
        print(f'Total Average PSNR: {float(table[:, 1].nanmean()):.3f}  images: {table.shape[0]}')
    sharding.barrier()
    return table


def _psnr_row(s, samples, x_orig_1):
    if samples.shape[0] == 0:
        return [float(s), float('nan'), 0.0]
    ps = torch.stack([K.psnr(samples[j:j + 1].contiguous(), x_orig_1)[0] for j in range(samples.shape[0])])
    return [float(s), float(ps.mean()), float(ps.std()) if ps.numel() > 1 else 0.0]


def main(argv=None):
    opt, _unknown = get_parser().parse_known_args(argv)
    if opt.algo == 'hmc_latent':
        raise NotImplementedError('--algo hmc_latent is served by main_sampling_latent.py (nhmc.cli.main_latent), as in the reference')
    if opt.algo != 'hmc':
        raise NotImplementedError(f"--algo {opt.algo}: this build serves the noise-space HMC path (--algo hmc) only")
    config, rank, world, device, op, seq, seq_next, images = _setup(opt, latent=False)
    size, ch = config['data']['image_size'], config['data']['channels']
    mc = dict(config['model'])
    mc.pop('var_type', None)
    model = unet.create_model(**mc).to(device).eval().requires_grad_(False)
    algo = plugin.HMC(model, op, opt.sigma_0)
    d = config['diffusion']
    b = torch.from_numpy(schedule.get_beta_schedule(d['beta_schedule'], beta_start=d['beta_start'], beta_end=d['beta_end'],
                                                    num_diffusion_timesteps=d['num_diffusion_timesteps'])).float().to(device)

    def probe():
        eng = sampler.LeapfrogEngine(algo.score, op, b, seq, seq_next, device)
        eng.decode_and_grad(torch.zeros(1, ch, size, size, device=device), torch.zeros(1, op.M, device=device))
    if world > 1 and rank != 0:
        sharding.barrier()                                               # rank 0 probes (and fills MIOpen's cache) first
    opt.score_chunk = auto_score_chunk(opt, device, probe, world)
    if world > 1 and rank == 0:
        sharding.barrier()
    if world > 1:
        # the first score-network call on a machine fills MIOpen's on-disk kernel cache (~1 min): one rank does it at
        # the shape the run will use, the others wait instead of racing through the same compiles
        if rank == 0:
            n0 = max(1, min(opt.chains, opt.score_chunk or opt.chains))
            xw = torch.zeros(n0, ch, size, size, device=device, requires_grad=True)
            with torch.enable_grad():
                out = model(xw, torch.full((n0,), 500.0, device=device))
            torch.autograd.grad(out, xw, torch.ones_like(out))
            torch.cuda.synchronize()
        sharding.barrier()
    rows = []
    for batch in image_batches(images.shape[0], rank, world, opt.chains):
        x_orig = images[batch[0]:batch[-1] + 1].to(device).contiguous()
        n = x_orig.shape[0]
        drawn = [draw_inputs(opt.seed, s, yk, opt.sigma_0, (ch, size, size)) for s, yk in zip(batch, op.H(x_orig))]
        y_0 = torch.stack([d_[0] for d_ in drawn]).contiguous()
        x = torch.stack([d_[1] for d_ in drawn]).contiguous()
        opt.chain_id0 = batch[0]
        out = run_with_oom_backoff(opt, lambda: sampler.hmc(x, n, b, seq, seq_next, algo, opt, y_0, op, x_orig))
        samples = out[None] if n == 1 else out                              # [n, 20, C, H, W]
        for k, s in enumerate(batch):
            rows.append(_psnr_row(s, samples[k], x_orig[k:k + 1]))
            if opt.save_images:
                sampler._save_png(samples[k].mean(0), os.path.join(opt.image_folder, f'{s}_mean.png'))
    return _report(rows, images.shape[0], rank, world, device)


def main_latent(argv=None):
    """main_sampling_latent.py:899-918 + `sample_image` (:351-560) for `--algo hmc_latent`: the latent model comes from
    configs/config_{dataset}_latent.yml + models/ldm/model.ckpt (:124-127), the operator acts on the decoded
    256 x 256 image, the sampler returns latents and `decode_first_stage` turns them into images (:477)."""
    opt, _unknown = get_parser(latent=True).parse_known_args(argv)
    if opt.algo != 'hmc_latent':
        raise NotImplementedError(f"--algo {opt.algo}: the latent entry serves --algo hmc_latent only")
    config, rank, world, device, op, seq, seq_next, images = _setup(opt, latent=True)
    model = ldm.create_latent_model(config['model'], ckpt='models/ldm/model.ckpt', quiet=rank != 0).to(device)
    algo = plugin.HMCLatent(model, op, opt.sigma_0)
    zc, zs = model.channels, model.image_size

    def probe():
        table = torch.cat([model.alphas_cumprod_prev[0:1], model.alphas_cumprod], dim=0)
        eng = sampler.LeapfrogEngine(algo.score, op, None, seq, seq_next, device, alpha_table=table,
                                     image_map=model.differentiable_decode_first_stage)
        eng.decode_and_grad(torch.zeros(1, zc, zs, zs, device=device), torch.zeros(1, op.M, device=device))
    opt.score_chunk = auto_score_chunk(opt, device, probe, world)
    if world > 1:
        sharding.barrier()
    rows = []
    for batch in image_batches(images.shape[0], rank, world, opt.chains):
        x_orig = images[batch[0]:batch[-1] + 1].to(device).contiguous()
        n = x_orig.shape[0]
        drawn = [draw_inputs(opt.seed, s, yk, opt.sigma_0, (zc, zs, zs)) for s, yk in zip(batch, op.H(x_orig))]
        y_0 = torch.stack([d_[0] for d_ in drawn]).contiguous()
        x = torch.stack([d_[1] for d_ in drawn]).contiguous()
        opt.chain_id0 = batch[0]
        out = run_with_oom_backoff(opt, lambda: sampler.hmc_latent(x, n, seq, seq_next, algo, opt, y_0, op, x_orig))
        per_chain = [out] if n == 1 else out                               # latents [<=10, C, h, w] per chain
        for k, s in enumerate(batch):
            imgs = model.decode_first_stage(per_chain[k]) if per_chain[k].shape[0] else per_chain[k]
            rows.append(_psnr_row(s, imgs, x_orig[k:k + 1]))
            if opt.save_images and imgs.shape[0]:
                sampler._save_png(imgs.mean(0), os.path.join(opt.image_folder, f'{s}_mean.png'))
    return _report(rows, images.shape[0], rank, world, device)


if __name__ == '__main__':
    main()
