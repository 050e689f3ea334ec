"""Host-side index maps between the flat parameter vector of a flow and the per-layer LDS weight images the
gfx950 kernels consume (layout documented in include/mentflow_hip.h and mentflow_amd/csrc/flow.hip).

image (floats), hidden width 64, ``nblk`` output blocks of 64 padded rows (RQS: one block per feature; affine: 1):
    [W0   : 64 x S0 ]  S0 = d | 1 (odd row stride)           input layer, natural [out][in]
    [b0   : 64      ]
    {[W_l : 64 x 65 ][b_l : 64]}  l = 1 .. L-1                hidden layers, natural [out][in], stride 65
    [Wout : nblk x 64 x 65]                                   last layer, ROWS PERMUTED (below)
    [bout : nblk x 64]

Masked-out weights and padding carry index -1 (the gather writes 0 there) — this is where zuko's
``mask * weight`` (MaskedLinear.forward) happens: once per optimizer step, not once per call.

Hidden-unit placement (mask sparsity).  zuko gives hidden unit u of every hidden layer the dependency class
c_u = 1 + (u mod (d-1)) (masks.py).  The images place the hidden units SORTED BY CLASS: sorted position j sits in
MFMA row  phys(j) = 32*((j>>1)>>4) + rowmap((j>>1)&15, j&1)  i.e. k-step s of the kernels covers sorted units
(2s, 2s+1).  With that placement the autoregressive masks are block-triangular in k-step space: an output block of
feature order o only needs the first ceil(cum[o]/2) k-steps (cum[c] = number of hidden units of class <= c), a hidden
output tile only the k-steps up to its largest class, and the kernels skip the MFMAs that would multiply zeros
(about half of them for d = 6).  Results equal the dense product up to fp32 summation order.

Row permutation of the last layer.  The kernels evaluate the spline of feature i in the two lanes (col, col+32) of
a particle; a lane half hh holds, in accumulator slot m (0..31), the MFMA row
    rho(hh, m) = 32*(m >> 4) + (m & 3) + 8*((m & 15) >> 2) + 4*hh.
RQS, K bins, KD0 = K // 2:   half 0: slots 0..K-1 = widths, DS..DS+KD0-1 = derivatives 0..KD0-1
                             half 1: slots 0..K-1 = heights, DS..      = derivatives KD0..K-2
                             DS = K for the compile-time instances (K = 8, 20), 21 for the run-time-bins instance
affine:                      half 0: slot i = shift_i;  half 1: slot i = scale_i      (single block)
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

HID = 64
WS = HID + 1


def rho(hh: int, m: int) -> int:
    return 32 * (m >> 4) + (m & 3) + 8 * ((m & 15) >> 2) + 4 * hh


def slot_of_row(r: int) -> Tuple[int, int]:
    """inverse of rho: physical row (0..63) -> (hh, slot m)."""
    rr = r & 31
    hh = (rr >> 2) & 1
    m = 16 * (r >> 5) + (rr & 3) + 4 * (rr >> 3)
    return hh, m


def hidden_placement(d: int, width: int = HID) -> np.ndarray:
    """phys[u] = image row of hidden unit u (units sorted by dependency class, two per k-step).

    width < 64 (``hidden_units`` below the kernels' width; mentflow/generate/build.py:36-38 takes it from the config): the
    image keeps its 64 rows and the CLASS SEGMENTS of the 64-wide layout — class c owns the sorted positions
    [cum64[c-1], cum64[c]) — and the narrower layer fills the first positions of every segment; the remaining rows stay
    empty (index -1: zero weights, zero bias, relu(0) = 0, gradients never read back).  Every mask-derived k-step bound the
    kernels compute for the 64-wide structure (make_sparsity in flow_kernels.inc) then still covers exactly the non-zero
    blocks, so the sparse forward, the fused backward and the activation hand-off run unchanged — at the cost of the 64-wide
    layer."""
    if not 1 <= width <= HID:
        raise NotImplementedError(f"hidden width {width}: the kernels hold a 64-wide conditioner")
    cls = 1 + (np.arange(width) % (d - 1))
    cum64 = np.concatenate([[0], class_counts(d, HID)[1:]]) if d > 1 else np.array([0, HID])   # cum64[c] = units of class <= c
    phys = np.empty(width, dtype=np.int64)
    for u in range(width):
        j = int(cum64[cls[u] - 1]) + u // (d - 1)          # segment start of the unit's class + its rank inside the class
        s_, half = j >> 1, j & 1
        phys[u] = 32 * (s_ >> 4) + rho(half, s_ & 15)
    return phys


def class_counts(d: int, width: int = HID) -> np.ndarray:
    """cum[c] = number of hidden units with dependency class <= c, c = 0..d-1."""
    cls = 1 + (np.arange(width) % (d - 1))
    return np.array([(cls <= c).sum() for c in range(d)], dtype=np.int64)


def image_layout(d: int, L: int, nblk: int) -> dict:
    S0 = d | 1
    g = {"S0": S0, "offW0": 0}
    g["offB0"] = HID * S0
    g["offWh"] = g["offB0"] + HID
    g["offW3"] = g["offWh"] + (L - 1) * (HID * WS + HID)
    g["offB3"] = g["offW3"] + nblk * HID * WS
    g["total"] = g["offB3"] + nblk * HID
    return g


def rqs_logical_param(hh: int, m: int, K: int, DS: int = None) -> int:
    """index (0..3K-2) of the spline parameter held in slot m of lane half hh, or -1 if the slot is unused.  DS = slot of
    the half's first derivative logit: K for the compile-time kernel instances, 21 for the run-time-bins instance
    (mf_flow_rqs_deriv_slot; slots K..DS-1 are unused there)."""
    DS = K if DS is None else DS
    KD0 = K // 2
    if m < K:
        return m if hh == 0 else K + m
    if m < DS:
        return -1
    j = m - DS
    if hh == 0:
        return 2 * K + j if j < KD0 else -1
    return 2 * K + KD0 + j if j < (K - 1 - KD0) else -1


def layer_image_index(d: int, L: int, kind: str, K: int, masks: Sequence[torch.Tensor], offsets: Sequence[int],
                      deriv_slot: int = None) -> np.ndarray:
    """int32 [image_floats]: index into the flat parameter vector (or -1).

    masks: bool [out,in] per linear layer (L+1 of them); offsets: flat offset of W_0, b_0, W_1, b_1, ... (2(L+1));
    deriv_slot: see rqs_logical_param (default: the compact layout of the compile-time instances).
    """
    total_per_feature = 3 * K - 1 if kind == "rqs" else 2
    nblk = d if kind == "rqs" else 1
    g = image_layout(d, L, nblk)
    idx = np.full(g["total"], -1, dtype=np.int64)
    width = int(masks[0].shape[0])                        # hidden units of the conditioner (<= 64: see hidden_placement)
    if any(int(m.shape[0]) != width for m in masks[:L]) or int(masks[L].shape[1]) != width:
        raise NotImplementedError("the hidden layers of a conditioner must share one width")
    phys = hidden_placement(d, width)
    m0 = masks[0].numpy()
    for u in range(width):
        for j in range(d):
            if m0[u, j]:
                idx[g["offW0"] + phys[u] * g["S0"] + j] = offsets[0] + u * d + j
        idx[g["offB0"] + phys[u]] = offsets[1] + u
    for l in range(1, L):
        ml = masks[l].numpy()
        base = g["offWh"] + (l - 1) * (HID * WS + HID)
        for u in range(width):
            for k in range(width):
                if ml[u, k]:
                    idx[base + phys[u] * WS + phys[k]] = offsets[2 * l] + u * width + k
            idx[base + HID * WS + phys[u]] = offsets[2 * l + 1] + u
    mo = masks[L].numpy()
    for blk in range(nblk):
        for r in range(HID):
            hh, m = slot_of_row(r)
            if kind == "rqs":
                t = rqs_logical_param(hh, m, K, deriv_slot)
                row = blk * total_per_feature + t if t >= 0 else -1
            else:
                row = (2 * m + hh) if m < d else -1       # feature m: (shift, scale) = rows 2m, 2m+1
            if row < 0:
                continue
            for k in range(width):
                if mo[row, k]:
                    idx[g["offW3"] + (blk * HID + r) * WS + phys[k]] = offsets[2 * L] + row * width + k
            idx[g["offB3"] + blk * HID + r] = offsets[2 * L + 1] + row
    return idx.astype(np.int32)


def invert_index(image_index: np.ndarray, numel: int) -> np.ndarray:
    """int32 [numel]: position in the (concatenated) image of every flat parameter, -1 if it is masked out."""
    inv = np.full(numel, -1, dtype=np.int64)
    pos = np.nonzero(image_index >= 0)[0]
    inv[image_index[pos]] = pos
    return inv.astype(np.int32)


# ------------------------------------------------------------------------------------------------------------------------
# Wide conditioners (hidden_units 65 .. 128 and / or 8 .. 16 features): mentflow_amd/csrc/flow_wide.hip.  The weights stay
# in global memory as 32 x 32 MFMA FRAGMENT blocks (forward and transposed copies); the gradient image is a separate,
# natural-order layout.  Hidden units are again placed sorted by dependency class, over the actual width.
WIDE_HT = 4
WIDE_HP = 32 * WIDE_HT
WIDE_DMAX = 16
WIDE_FB = 1024


def wide_placement(d: int, width: int) -> np.ndarray:
    """phys[u] = physical row (0 .. 127) of hidden unit u: units sorted by dependency class c_u = 1 + (u mod (d-1)), sorted
    position j in MFMA row 32*((j>>1)>>4) + rowmap((j>>1)&15, j&1) — so that sorted positions 32t .. 32t+31 are hidden tile t and
    the autoregressive masks are block-triangular over tiles (make_wide_sp in flow_wide.hip computes the same counts)."""
    if not 1 <= width <= WIDE_HP:
        raise NotImplementedError(f"hidden width {width}: the wide kernels hold up to {WIDE_HP} units")
    if d > 1:
        cls = 1 + (np.arange(width) % (d - 1))
        cum = np.array([(cls <= c).sum() for c in range(d)], dtype=np.int64)
        j = cum[cls - 1] + np.arange(width) // (d - 1)
    else:
        j = np.arange(width)
    s_, half = j >> 1, j & 1
    return (32 * (s_ >> 4) + (s_ & 3) + 8 * ((s_ & 15) >> 2) + 4 * half).astype(np.int64)


def wide_layout(L: int, nblk: int) -> dict:
    g = {"offW0F": 0}
    g["offB0"] = WIDE_HT * 512
    g["offW0T"] = g["offB0"] + WIDE_HT * 32
    g["offH"] = g["offW0T"] + WIDE_HT * WIDE_FB
    g["strideH"] = 2 * WIDE_HT * WIDE_HT * WIDE_FB + WIDE_HT * 32
    g["off3"] = g["offH"] + (L - 1) * g["strideH"]
    g["stride3"] = 2 * WIDE_HT * WIDE_FB + WIDE_HT * 2 * WIDE_FB + 64
    g["total"] = g["off3"] + nblk * g["stride3"]
    return g


def wide_grad_layout(L: int, nblk: int) -> dict:
    g = {"offW0": 0, "offB0": WIDE_HP * WIDE_DMAX}
    g["offH"] = g["offB0"] + WIDE_HP
    g["strideH"] = WIDE_HP * WIDE_HP + WIDE_HP
    g["off3"] = g["offH"] + (L - 1) * g["strideH"]
    g["stride3"] = 64 * WIDE_HP + 64
    g["total"] = g["off3"] + nblk * g["stride3"]
    return g


_LANE = np.arange(64)
_COL, _HH = _LANE & 31, _LANE >> 5
_G4, _J4 = np.meshgrid(np.arange(4), np.arange(4), indexing="ij")              # [g, j]
_KROW = ((_G4 * 4 + _J4) & 3)[:, None, :] + 8 * ((_G4 * 4 + _J4) >> 2)[:, None, :] + 4 * _HH[None, :, None]   # rowmap(4g+j, hh): [g, lane, j]
_COLB = np.broadcast_to(_COL[None, :, None], (4, 64, 4))


def _frag_f(nat: np.ndarray, rt: int, it: int) -> np.ndarray:
    """forward fragment block (rt, it): (g, lane, j) -> nat[32 rt + col][32 it + rowmap(4g + j, hh)]"""
    return nat[32 * rt + _COLB, 32 * it + _KROW].reshape(-1)


def _frag_t(nat: np.ndarray, it: int, kt: int) -> np.ndarray:
    """transposed fragment block (it, kt): (g, lane, j) -> nat[32 kt + rowmap(4g + j, hh)][32 it + col]"""
    return nat[32 * kt + _KROW, 32 * it + _COLB].reshape(-1)


def _frag_bias(nb: np.ndarray, tiles: int) -> np.ndarray:
    """(rt, hh, r) -> nb[32 rt + rowmap(r, hh)]"""
    r = np.arange(16)
    rows = np.stack([32 * rt + (r & 3) + 8 * (r >> 2) + 4 * hh for rt in range(tiles) for hh in range(2)])
    return nb[rows].reshape(-1)


def wide_image_index(d: int, L: int, kind: str, K: int, masks: Sequence[torch.Tensor], offsets: Sequence[int],
                     deriv_slot: int = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(image_index int32 [wide image floats] -> flat parameter index or -1,
        grad_param int64 [m], grad_pos int64 [m]: parameter grad_param[i] receives position grad_pos[i] of the gradient image).

    Arguments as layer_image_index.  Every unmasked parameter appears in the weight image twice (forward and transposed
    fragments; biases once) and in the gradient image once."""
    if d > WIDE_DMAX:
        raise NotImplementedError(f"the wide flow kernels take up to {WIDE_DMAX} features (got {d})")
    total_per_feature = 3 * K - 1 if kind == "rqs" else 2
    nblk = d if kind == "rqs" else 1
    width = int(masks[0].shape[0])
    if any(int(m.shape[0]) != width for m in masks[:L]) or int(masks[L].shape[1]) != width:
        raise NotImplementedError("the hidden layers of a conditioner must share one width")
    phys = wide_placement(d, width)
    g, gg = wide_layout(L, nblk), wide_grad_layout(L, nblk)
    img = np.full(g["total"], -1, dtype=np.int64)
    gparam: List[np.ndarray] = []
    gpos: List[np.ndarray] = []

    def grad_entries(nat: np.ndarray, off: int, stride: int) -> None:
        r, c = np.nonzero(nat >= 0)
        gparam.append(nat[r, c])
        gpos.append(off + r * stride + c)

    def grad_bias(nb: np.ndarray, off: int) -> None:
        r = np.nonzero(nb >= 0)[0]
        gparam.append(nb[r])
        gpos.append(off + r)

    # input layer
    nat0 = np.full((WIDE_HP, 32), -1, dtype=np.int64)             # 32 columns: the transposed fragments index col = 0 .. 31
    m0 = masks[0].numpy().astype(bool)
    u, j = np.nonzero(m0)
    nat0[phys[u], j] = offsets[0] + u * d + j
    nb0 = np.full(WIDE_HP, -1, dtype=np.int64)
    nb0[phys] = offsets[1] + np.arange(width)
    for rt in range(WIDE_HT):
        for gx in range(2):
            jj = np.arange(4)
            colsx = 2 * (4 * gx + jj)[None, :] + _HH[:, None]                      # [lane, j] -> input feature 2 (4 gx + j) + hh
            img[g["offW0F"] + (rt * 2 + gx) * 256:][:256] = nat0[32 * rt + _COL[:, None], colsx].reshape(-1)
        img[g["offW0T"] + rt * WIDE_FB:][:WIDE_FB] = _frag_t(nat0, 0, rt)
    img[g["offB0"]:][:WIDE_HT * 32] = _frag_bias(nb0, WIDE_HT)
    grad_entries(nat0[:, :WIDE_DMAX], gg["offW0"], WIDE_DMAX)
    grad_bias(nb0, gg["offB0"])
    # hidden layers
    for l in range(1, L):
        ml = masks[l].numpy().astype(bool)
        nat = np.full((WIDE_HP, WIDE_HP), -1, dtype=np.int64)
        u, k = np.nonzero(ml)
        nat[phys[u], phys[k]] = offsets[2 * l] + u * width + k
        nb = np.full(WIDE_HP, -1, dtype=np.int64)
        nb[phys] = offsets[2 * l + 1] + np.arange(width)
        base = g["offH"] + (l - 1) * g["strideH"]
        for a in range(WIDE_HT):
            for b in range(WIDE_HT):
                img[base + (a * WIDE_HT + b) * WIDE_FB:][:WIDE_FB] = _frag_f(nat, a, b)
                img[base + WIDE_HT * WIDE_HT * WIDE_FB + (a * WIDE_HT + b) * WIDE_FB:][:WIDE_FB] = _frag_t(nat, a, b)
        img[base + 2 * WIDE_HT * WIDE_HT * WIDE_FB:][:WIDE_HT * 32] = _frag_bias(nb, WIDE_HT)
        gbase = gg["offH"] + (l - 1) * gg["strideH"]
        grad_entries(nat, gbase, WIDE_HP)
        grad_bias(nb, gbase + WIDE_HP * WIDE_HP)
    # output blocks
    mo = masks[L].numpy().astype(bool)
    for blk in range(nblk):
        nat = np.full((64, WIDE_HP), -1, dtype=np.int64)
        nb = np.full(64, -1, dtype=np.int64)
        for r in range(64):
            hh, m = slot_of_row(r)
            if kind == "rqs":
                t = rqs_logical_param(hh, m, K, deriv_slot)
                row = blk * total_per_feature + t if t >= 0 else -1
            else:
                row = (2 * m + hh) if m < d else -1
            if row < 0:
                continue
            k = np.nonzero(mo[row])[0]
            nat[r, phys[k]] = offsets[2 * L] + row * width + k
            nb[r] = offsets[2 * L + 1] + row
        base = g["off3"] + blk * g["stride3"]
        for rt in range(2):
            for it in range(WIDE_HT):
                img[base + (rt * WIDE_HT + it) * WIDE_FB:][:WIDE_FB] = _frag_f(nat, rt, it)
                img[base + 2 * WIDE_HT * WIDE_FB + (it * 2 + rt) * WIDE_FB:][:WIDE_FB] = _frag_t(nat, it, rt)
        img[base + 4 * WIDE_HT * WIDE_FB:][:64] = _frag_bias(nb, 2)
        gbase = gg["off3"] + blk * gg["stride3"]
        grad_entries(nat, gbase, WIDE_HP)
        grad_bias(nb, gbase + 64 * WIDE_HP)
    return img.astype(np.int32), np.concatenate(gparam), np.concatenate(gpos)
