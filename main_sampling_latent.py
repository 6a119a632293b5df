#!/usr/bin/env python3
"""Entry point with the reference's name and flags (main_sampling_latent.py); everything lives in
noise-space-hmc_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from nhmc.cli import main_latent  # noqa: E402

if __name__ == '__main__':
    main_latent()
