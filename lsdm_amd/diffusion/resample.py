"""Timestep samplers of the training loop (reference diffusion/resample.py:8-62).

``run/train_sdm.py:33-34`` creates the ``'uniform'`` sampler; the loss-aware samplers of the reference (:64-154) are never
constructed by LSDM (and their synchronisation code is dead), so only the uniform one exists here."""
import numpy as np
import torch as th


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")


class UniformSampler:
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights

    def sample(self, batch_size, device):
        """Importance-samples timesteps with numpy's global generator, as the reference does (resample.py:50-57):
        -> (timesteps int64 (B,), weights float32 (B,))."""
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        return th.from_numpy(idx).long().to(device), th.from_numpy(1 / (len(p) * p[idx])).float().to(device)
