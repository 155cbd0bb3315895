"""Decoder of the Middlebury flow colour code the reference's celiu/ directory writes its results in (celiu/flowToColor.m,
celiu/computeColor.m; test infrastructure).  celiu/output/car_flow.jpg -- the one OUTPUT of the named ground truth (Ce Liu's
Coarse2FineTwoFrames on car1.jpg -> car2.jpg, celiu/demoflow.m:43-47) that the reference holds -- is such an image:
    u, v = flow / (max |flow| + eps)                               flowToColor.m:77-78 (the maximum is printed, not stored)
    a = atan2(-v, -u) / pi;  fk = (a + 1) / 2 * (ncols - 1) + 1    computeColor.m:42-44: position on a wheel of ncols = 55 colours
    col = (1 - f) wheel[k0] + f wheel[k1]                          :46-56
    col = 1 - rad (1 - col)   where rad = |(u, v)| <= 1            :58-59: saturation grows with the magnitude
so a pixel gives back the flow's DIRECTION (hue) and its magnitude RELATIVE to the field's maximum (1 - the smallest channel: every
wheel colour has one channel at zero); the absolute scale is not recoverable from the file."""
import numpy as np


def make_colorwheel():
    """computeColor.m:70-115: RY 15, YG 6, GC 4, CB 11, BM 13, MR 6 steps; r, g, b in 0..255"""
    RY, YG, GC, CB, BM, MR = 15, 6, 4, 11, 13, 6
    w = np.zeros((RY + YG + GC + CB + BM + MR, 3))
    c = 0
    w[0:RY, 0] = 255; w[0:RY, 1] = np.floor(255 * np.arange(RY) / RY); c += RY
    w[c : c + YG, 0] = 255 - np.floor(255 * np.arange(YG) / YG); w[c : c + YG, 1] = 255; c += YG
    w[c : c + GC, 1] = 255; w[c : c + GC, 2] = np.floor(255 * np.arange(GC) / GC); c += GC
    w[c : c + CB, 1] = 255 - np.floor(255 * np.arange(CB) / CB); w[c : c + CB, 2] = 255; c += CB
    w[c : c + BM, 2] = 255; w[c : c + BM, 0] = np.floor(255 * np.arange(BM) / BM); c += BM
    w[c : c + MR, 2] = 255 - np.floor(255 * np.arange(MR) / MR); w[c : c + MR, 0] = 255
    return w


def encode(u, v):
    """computeColor.m:32-66 for |(u, v)| <= 1 (forward direction: used to test the decoder)"""
    wheel = make_colorwheel()
    n = wheel.shape[0]
    rad = np.sqrt(u * u + v * v)
    a = np.arctan2(-v, -u) / np.pi
    fk = (a + 1) / 2 * (n - 1)                  # 0-based
    k0 = np.floor(fk).astype(int)
    k1 = (k0 + 1) % n
    f = (fk - k0)[..., None]
    col = (1 - f) * wheel[k0] / 255 + f * wheel[k1] / 255
    col = 1 - rad[..., None] * (1 - col)
    return np.floor(255 * col).astype(np.uint8)


def decode(img, sub=8):
    """img [H][W][3] uint8 -> (ux, uy, rad): unit direction of the flow (x right, y down) and its magnitude relative to the field's
    maximum.  The hue is matched against the wheel sampled `sub` times between neighbouring colours (nearest in RGB)."""
    wheel = make_colorwheel() / 255
    n = wheel.shape[0]
    col = img.astype(np.float64) / 255
    rad = 1 - col.min(axis=2)
    pure = 1 - (1 - col) / np.maximum(rad, 1e-3)[..., None]
    fk = np.arange(0, (n - 1) * sub + 1) / sub                       # positions 0 .. n-1 on the wheel
    k0 = np.floor(fk).astype(int)
    k1 = np.minimum(k0 + 1, n - 1)
    f = (fk - k0)[:, None]
    samples = (1 - f) * wheel[k0] + f * wheel[k1]                    # [S][3]
    d = ((pure[..., None, :] - samples[None, None]) ** 2).sum(axis=3)
    best = fk[d.argmin(axis=2)]
    a = best / (n - 1) * 2 - 1                                       # atan2(-v, -u) / pi
    return -np.cos(a * np.pi), -np.sin(a * np.pi), rad
