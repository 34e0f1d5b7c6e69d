timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "perturbed_latent or train_mode_masks or training_step" 2>&1 | tail -12
