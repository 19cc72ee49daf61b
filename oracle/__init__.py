"""CPU oracle for the denoising hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``transformer_latent_diffusion_amd/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.
"""
