#!/bin/bash
# Full-size (256x256, FFHQ U-Net architecture, random init) smoke of the command line for every degradation on the hot
# path: a short annealing + sampling run each.  Usage on the GPU box: bash tools/cli_sweep.sh > gpurun_out/cli_sweep.log
set -u
OUT=${OUT:-/tmp/nhmc_cli_sweep}
for deg in inpaint_random inpaint_box sr4 sr16 sr_bicubic4 deblur_aniso deblur_gauss color cs4; do
  sig=0.05; [ $deg = deblur_aniso ] && sig=0.01
  echo "== $deg"
  timeout -k 10 300 python main_sampling.py --dataset ffhq --algo hmc --timesteps 3 --deg $deg --sigma_0 $sig -i $OUT/$deg \
      --tau 0.25 --epsilon 0.05 --synthetic 2 --chains 2 --philox --hmc_epochs 3 --hmc_sampling 2 2>&1 | grep -v amdgpu.ids | tail -4
  echo "rc=$?"
done
echo "== deblur_aniso --spectral_projected"
timeout -k 10 300 python main_sampling.py --dataset ffhq --algo hmc --timesteps 3 --deg deblur_aniso --sigma_0 0.01 -i $OUT/aniso_proj \
    --tau 0.25 --epsilon 0.05 --synthetic 2 --chains 2 --philox --hmc_epochs 3 --hmc_sampling 2 --spectral_projected 2>&1 | grep -v amdgpu.ids | tail -4
echo "== hmc_latent"
timeout -k 10 300 python main_sampling_latent.py --dataset ffhq --algo hmc_latent --timesteps 3 --deg inpaint_random --sigma_0 0.05 -i $OUT/latent \
    --tau 0.3 --epsilon 0.1 --synthetic 2 --chains 2 --philox --hmc_epochs 3 --hmc_sampling 2 2>&1 | grep -v amdgpu.ids | tail -4
