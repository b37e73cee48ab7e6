"""Synthetic bouncing-ball video (no dataset ships with the reference: kvae/train/config.yaml:8 points at a
user's laptop).  Binary 32x32 frames of one disc of radius 3 px bouncing elastically in [3,28]^2
(SURVEY.md §8d): position ~ U[6,25]^2, heading ~ U[0,2pi), speed ~ U[0.5,2] px/step."""
import numpy as np
import torch


def bouncing_ball(B, T, seed, size=32, radius=3.0):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(6, size - 7, size=(B, 2))
    ang = rng.uniform(0, 2 * np.pi, size=B)
    spd = rng.uniform(0.5, 2.0, size=B)
    vel = np.stack([np.cos(ang), np.sin(ang)], 1) * spd[:, None]
    yy, xx = np.mgrid[0:size, 0:size]
    lo_w, hi_w = radius, size - 1 - radius
    frames = np.zeros((B, T, 1, size, size), np.uint8)
    for t in range(T):
        d2 = (xx[None] - pos[:, 0, None, None]) ** 2 + (yy[None] - pos[:, 1, None, None]) ** 2
        frames[:, t, 0] = d2 <= radius * radius
        pos = pos + vel
        for d in range(2):
            lo, hi = pos[:, d] < lo_w, pos[:, d] > hi_w
            pos[lo, d] = 2 * lo_w - pos[lo, d]
            pos[hi, d] = 2 * hi_w - pos[hi, d]
            vel[lo | hi, d] *= -1
    return torch.from_numpy(frames)
