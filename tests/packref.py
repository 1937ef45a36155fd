"""Host-side (numpy) statement of the packed weight layouts consumed by rtg_conv1d — TEST INFRASTRUCTURE.
The product packs on the GPU (rtg_weights_pack); tests compare that kernel with this file and use it to drive
rtg_conv1d directly."""
import numpy as np

CK = 16


def pack_logical(W, TM):
    """W: [G, Mg, Cg, K] logical operator -> flat packed buffer [g][mt][cc][tap][cp][kk][m] (zero padded)."""
    G, Mg, Cg, K = W.shape
    KK = 64 // TM
    n_mt, n_cc = -(-Mg // TM), -(-Cg // CK)
    P = np.zeros((G, n_mt * TM, n_cc * CK, K), dtype=np.float32)
    P[:, :Mg, :Cg] = W
    P = P.reshape(G, n_mt, TM, n_cc, CK // KK, KK, K)          # g, mt, m, cc, cp, kk, tap
    return np.ascontiguousarray(P.transpose(0, 1, 3, 6, 4, 5, 2)).reshape(-1)


def pack_logical_tapmajor(W, TM):
    """tap-major K order: [g][mt][k-step group][cp][kk][m], k-step = (channel, group of KK consecutive taps)."""
    G, Mg, Cg, K = W.shape
    KK = 64 // TM
    CPN = CK // KK
    TG = -(-K // KK)
    n_mt, n_grp = -(-Mg // TM), -(-(Cg * TG) // CPN)
    P = np.zeros((G, n_mt, n_grp, CPN, KK, TM), dtype=np.float32)
    for c in range(Cg):
        for tg in range(TG):
            ks = c * TG + tg
            for kk in range(KK):
                j = tg * KK + kk
                if j < K:
                    for mt in range(n_mt):
                        rows = W[:, mt * TM:(mt + 1) * TM, c, j]
                        P[:, mt, ks // CPN, ks % CPN, kk, :rows.shape[1]] = rows
    return P.reshape(-1)


def logical_fwd(w, groups):
    """torch Conv1d weight [C_out, Cg, K] -> [G, Mg, Cg, K]."""
    C_out, Cg, K = w.shape
    return w.reshape(groups, C_out // groups, Cg, K)


def logical_dgrad_s1(w, groups):
    """conv backward-data, stride 1: rows = input channels, reduce over output channels, taps flipped."""
    C_out, Cg, K = w.shape
    Mg = C_out // groups
    return np.ascontiguousarray(w.reshape(groups, Mg, Cg, K).transpose(0, 2, 1, 3)[..., ::-1])


def logical_dgrad_poly(w, groups, s):
    """conv backward-data, stride s (dilation 1) as a polyphase stride-1 conv: rows = (c, phase r)."""
    C_out, Cg, K = w.shape
    Mg = C_out // groups
    nt = -(-K // s)
    wg = w.reshape(groups, Mg, Cg, K)
    W = np.zeros((groups, Cg * s, Mg, nt), dtype=np.float32)
    for r in range(s):
        for ip in range(nt):
            j = r + (nt - 1 - ip) * s
            if j < K:
                W[:, r::s, :, ip] = wg[:, :, :, j].transpose(0, 2, 1)
    return W


def logical_convT_poly(w, s):
    """ConvTranspose1d weight [C_in, C_out, K] forward as polyphase conv: rows = (co, phase r), reduce over ci."""
    C_in, C_out, K = w.shape
    nt = -(-K // s)
    W = np.zeros((1, C_out * s, C_in, nt), dtype=np.float32)
    for r in range(s):
        for ip in range(nt):
            j = r + (nt - 1 - ip) * s
            if j < K:
                W[0, r::s, :, ip] = w[:, :, j].T
    return W


def logical_convT_dgrad(w):
    """ConvTranspose1d backward-data = strided conv of dy: rows = ci, reduce over co."""
    return w[None].copy()


def pack_frag16(W):
    """W: [1, Mg, Cg, K] logical operator -> the 16-byte-fragment image of rtg_dconv.hip (RtgPackJob.frag16):
    [16-row tile][chunk][tap][kgrp 4][m 16][kq 4] with channel = 16 * chunk + 4 * kq + kgrp, zero padded."""
    G, Mg, Cg, K = W.shape
    assert G == 1
    n_mt, n_cc = -(-Mg // 16), -(-Cg // CK)
    P = np.zeros((n_mt * 16, n_cc * CK, K), dtype=np.float32)
    P[:Mg, :Cg] = W[0]
    P = P.reshape(n_mt, 16, n_cc, 4, 4, K)                       # mt, m, cc, kq, kgrp, tap
    return np.ascontiguousarray(P.transpose(0, 2, 5, 4, 1, 3)).reshape(-1)


def bf16_bits(a):
    """fp32 array -> bf16 bit patterns (uint16), round to nearest even"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def pack_frag16_bf16(W):
    """W: [1, Mg, Cg, K] -> the bf16 fragment image of rtg_dconv.hip (RtgPackJob.frag16 with .bf16), as float32 words:
    [16-row tile][32-channel chunk][tap][kgrp 4][m 16][8 bf16] with channel = 32 * chunk + 8 * kgrp + element."""
    G, Mg, Cg, K = W.shape
    assert G == 1
    n_mt, n_cc = -(-Mg // 16), -(-Cg // 32)
    P = np.zeros((n_mt * 16, n_cc * 32, K), dtype=np.float32)
    P[:Mg, :Cg] = W[0]
    P = P.reshape(n_mt, 16, n_cc, 4, 8, K)                       # mt, m, cc, kgrp, el, tap
    bits = bf16_bits(np.ascontiguousarray(P.transpose(0, 2, 5, 3, 1, 4)))
    return bits.reshape(-1).view(np.float32)
