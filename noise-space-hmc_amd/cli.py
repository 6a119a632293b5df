"""Command line with the reference's flags (`get_parser`, main_sampling.py:923-1015) for the HMC path.

    python main_sampling.py --dataset ffhq --algo hmc --timesteps 3 --deg inpaint_random --sigma_0 0.05 \
        -i exp/samples/ffhq/inpaint_random/hmc --tau 1.0 --epsilon 0.05          (README.md:79 of the reference)

Same flag names and defaults; unknown flags are ignored as in the reference (`parse_known_args`, :1031).  Only
`--algo hmc` and the degradations on the hot path are served; anything else raises `NotImplementedError`
(the reference's own error for an unknown algo, :256-257).  Additions, all optional: `--chains` images are
sampled in parallel as independent chains (the reference is batch-1), `--score_chunk` bounds the U-Net batch,
`--synthetic K` uses K synthetic images when no dataset folder is present, `--philox` switches the noise to the
shard-invariant counter-based generator.  Under torchrun the images are sharded over ranks and the per-image
metrics are gathered once at the end (RCCL).
"""
import argparse
import glob
import math
import os
import random

import numpy as np
import torch

from . import operators, plugin, sampler, schedule, sharding, unet
from . import kernels as K


def get_parser():
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
    p.add_argument('--epsilon', type=float, default=0.05, help='Epsilon for HMC')
    p.add_argument('--sigma_y', type=float, default=0.5, help='sigma_y for HMC (measurement noise)')
    p.add_argument('--m', type=float, default=1.0, help='Mass Matrix Variance')
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
    p.add_argument('--score_chunk', type=int, default=16)
    p.add_argument('--synthetic', type=int, default=0, help='use this many synthetic images')
    p.add_argument('--philox', action='store_true', help='counter-based, shard-invariant noise')
    p.add_argument('--save_images', action='store_true')
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


def load_config(dataset):
    path = f'configs/config_{dataset}.yml'
    if os.path.exists(path):
        import yaml
        with open(path) as f:
            return yaml.safe_load(f)
    if dataset == 'ffhq':
        return FFHQ_DEFAULTS
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


def main(argv=None):
    opt, _unknown = get_parser().parse_known_args(argv)
    if opt.algo != 'hmc':
        raise NotImplementedError(f"--algo {opt.algo}: this build serves the noise-space HMC path (--algo hmc) only")
    config = load_config(opt.dataset)
    rank, local_rank, world = sharding.init_process_group()
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(opt.seed)
    np.random.seed(opt.seed)
    random.seed(opt.seed)

    size, ch = config['data']['image_size'], config['data']['channels']
    op = operators.build_operator(opt.deg, ch, size, device)               # mask = first torch RNG draw, as in the reference
    opt.sigma_0 = 2 * opt.sigma_0                                           # [-1,1] scaling, main_sampling.py:348
    mc = dict(config['model'])
    mc.pop('var_type', None)
    model = unet.create_model(**mc).to(device).eval().requires_grad_(False)
    algo = plugin.HMC(model, op, opt.sigma_0)
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
    d = config['diffusion']
    b = torch.from_numpy(schedule.get_beta_schedule(d['beta_schedule'], beta_start=d['beta_start'], beta_end=d['beta_end'],
                                                    num_diffusion_timesteps=d['num_diffusion_timesteps'])).float().to(device)
    skip = opt.num_timesteps // (opt.timesteps + 1)                          # main_sampling.py:469-471
    seq = list(range(skip, opt.num_timesteps, skip))
    seq_next = [-1] + seq[:-1]

    images = load_images(os.path.join(opt.exp, 'datasets', opt.dataset), size, opt.subset_start, opt.subset_end,
                         opt.synthetic, opt.seed)
    lo, hi = sharding.chain_range(images.shape[0], rank, world)
    opt.quiet = opt.chains > 1 or rank != 0
    opt.progress_every = 10 if rank == 0 else 0                           # stderr heartbeat for long quiet runs
    if opt.philox:
        opt.philox_seed = opt.seed
    rows = []
    for s in range(lo, hi, opt.chains):
        x_orig = images[s:min(hi, s + opt.chains)].to(device).contiguous()
        n = x_orig.shape[0]
        y_0 = op.H(x_orig)
        y_0 = y_0 + opt.sigma_0 * torch.randn_like(y_0)                     # main_sampling.py:447-448
        x = torch.randn(n, ch, size, size, device=device)
        opt.chain_id0, opt.score_chunk = s, opt.score_chunk
        out = sampler.hmc(x, n, b, seq, seq_next, algo, opt, y_0, op, x_orig)
        samples = out[None] if n == 1 else out                              # [n, 20, C, H, W]
        for k in range(n):
            ps = torch.stack([K.psnr(samples[k, j:j + 1].contiguous(), x_orig[k:k + 1])[0] for j in range(samples.shape[1])])
            rows.append([float(s + k), float(ps.mean()), float(ps.std()) if ps.numel() > 1 else 0.0])
            if opt.save_images:
                sampler._save_png(samples[k].mean(0), os.path.join(opt.image_folder, f'{s + k}_mean.png'))
    local = torch.tensor(rows, dtype=torch.float32, device=device).reshape(-1, 3)
    table = sharding.gather_chains(local, images.shape[0], rank, world).cpu()
    if rank == 0:
        for idx, mean, std in table.tolist():
            print(f'image {int(idx)}: PSNR {mean:.3f} (std over samples {std:.4f})')
        print(f'Total Average PSNR: {float(table[:, 1].mean()):.3f}  images: {table.shape[0]}')
    sharding.barrier()
    return table


if __name__ == '__main__':
    main()
