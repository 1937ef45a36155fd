// rtg_weights.hip — the weight bank: old-style weight norm (w = g * v / ||v||, norm over all dims but 0) for every
// layer of a model in ONE launch per stage, instead of the reference's per-layer hook (torch.nn.utils.weight_norm;
// 435 forward + 425 backward tiny launches per step).
//   rtg_weightnorm_scales    scale[r] = g[r] / ||v[r,:]||  and 1/||v[r,:]||
//   rtg_weights_pack         gathers v * scale into the MFMA-fragment layouts rtg_conv1d reads (forward, backward-data,
//                            polyphase transposed): [g][m-tile][c-chunk][tap][k-step][kk][m], zero padded
//   rtg_weightnorm_backward  sums the split partials of rtg_conv1d_wgrad in fixed order and applies the weight-norm
//                            chain rule, accumulating into the flat gradient buffer
#include "rtg_common.h"

namespace {

__global__ __launch_bounds__(RTG_THREADS) void wn_scales_kernel(const RtgNormJob* jobs, const float* params,
                                                                 float* scales) {
  __shared__ float red[4];
  const RtgNormJob j = jobs[blockIdx.y];
  for (int r = blockIdx.x; r < j.rows; r += gridDim.x) {
    const float* v = params + j.v_off + (size_t)r * j.inner;
    float ss = 0.f;
    for (int i = threadIdx.x; i < j.inner; i += RTG_THREADS) ss += v[i] * v[i];
    ss = rtg_block_sum(ss, red);
    if (threadIdx.x == 0) {
      const float n = sqrtf(ss);
      scales[j.scale_off + r] = params[j.g_off + r] / n;
      scales[j.scale_off + j.rows + r] = 1.f / n;
    }
  }
}

// RtgPackJob.src_T: index of the operator's element (ci, kernel row r, tap t) — `sin` in the operator's order [ci][KH][src_K]
// — in the source tensor, whose kernel axes are the other way round: [ci][src_K][KH]
__device__ __forceinline__ int pack_src_index(const RtgPackJob& j, int sin) {
  if (!j.src_T) return sin;
  const int cr = sin / j.src_K, t = sin - cr * j.src_K;
  const int ci = cr / j.KH, r = cr - ci * j.KH;
  return (ci * j.src_K + t) * j.KH + r;
}

// the effective weight g*v/||v|| that packed row `row` (of group g) applies to packed channel c at packed tap `tap`
__device__ __forceinline__ float pack_logical(const RtgPackJob& j, const float* params, const float* scales, int g,
                                            int row, int c, int tap) {
  const int inner = j.src_inner_c * j.src_K;
  float val = 0.f;
  if (row < j.Mg && c < j.Cg) {
    long long srow = -1, sin = 0;
    if (j.mode == RTG_PACK_FWD) {
      srow = (long long)g * j.Mg + row;
      sin = (long long)c * j.src_K + tap;
      if (j.kh_major) {        // packed channel (kernel row, ci) <- the operator's [ci][KH][src_K]
        const int n_ci = j.Cg / j.KH;
        const int kh = c / n_ci, ci = c - kh * n_ci;
        sin = ((long long)ci * j.KH + kh) * j.src_K + tap;
      }
    } else if (j.mode == RTG_PACK_DGRAD_S1) {
      srow = (long long)g * j.Cg + c;
      sin = (long long)row * j.src_K + (j.src_K - 1 - tap);
    } else if (j.mode == RTG_PACK_DGRAD_2D) {
      // source [C_out][C_in][KH][src_K]; packed rows (ci, phase r), packed channels (kh, co) — kernel row major, so
      // that a block of one row residue class walks whole chunk ranges (rtg_conv1d_kernel.h) —, taps along W
      const int ch = row / j.S, r = row - ch * j.S;
      const int n_co = j.Cg / j.KH;
      const int kh = c / n_co, co = c - kh * n_co;
      const int jj = r + (j.K - 1 - tap) * j.S;
      if (jj < j.src_K) {
        srow = co;
        sin = ((long long)ch * j.KH + kh) * j.src_K + jj;
      }
    } else {
      const int ch = row / j.S, r = row - ch * j.S;
      const int jj = r + (j.K - 1 - tap) * j.S;
      if (jj < j.src_K) {
        if (j.mode == RTG_PACK_DGRAD_POLY) {
          srow = (long long)g * j.Cg + c;
          sin = (long long)ch * j.src_K + jj;
        } else {   // RTG_PACK_CONVT_POLY (groups == 1): source [C_in][C_out][K]
          srow = c;
          sin = (long long)ch * j.src_K + jj;
        }
      }
    }
    if (srow >= 0) val = params[j.v_off + srow * inner + pack_src_index(j, (int)sin)] * scales[j.scale_off + srow];
  }
  return val;
}

// Every fp32 image is a sequence of 256-float (frag16) or 16 * tile_m-float steps, one per (row tile, chunk, tap) — or per
// (row tile, k-step group) in the tap-major order — whose inner index is a power of two.  A wave packs one whole step per
// iteration: the step's coordinates (three divisions by run-time values) are computed once per 256 / 512 elements, an
// element splits its index within the step with shifts and finds its source in 32-bit arithmetic.  (Round 1: 64-bit
// divisions per element were most of this kernel's time; round 2: 32-bit ones still 3/4 of it, ~110 instructions per
// element at 1.2 TB/s.)
// what a job's slabs look like (shared by the kernel and rtg_pack_job_blocks): rows of a row tile, source rows and run
// length of a slab, whether the slab fits the staging buffer
constexpr int kPackSlab = 32 * 241;                                          // source rows x (run + 1 float of padding)
struct PackGeom { int RT, n_rt, n_cc, nrow, run; bool staged; };
__host__ __device__ inline PackGeom pack_geom(const RtgPackJob& j) {
  PackGeom p;
  p.RT = j.frag16 ? 16 : j.tile_m;
  p.n_rt = (j.Mg + p.RT - 1) / p.RT;
  p.n_cc = (j.Cg + RTG_CK - 1) / RTG_CK;
  const int S = j.S > 0 ? j.S : 1;
  const int chs = (j.mode == RTG_PACK_DGRAD_POLY || j.mode == RTG_PACK_CONVT_POLY) ? (p.RT - 1) / S + 2 : 0;
  p.run = j.mode == RTG_PACK_FWD ? RTG_CK * j.src_K : (j.mode == RTG_PACK_DGRAD_S1 ? p.RT * j.src_K : chs * j.src_K);
  p.nrow = j.mode == RTG_PACK_FWD ? p.RT : RTG_CK;
  p.staged = !j.bf16 && !j.tap_major && !j.src_T && !j.kh_major && j.mode != RTG_PACK_DGRAD_2D && j.mode < RTG_PACK_GCONV_FWD &&
             p.nrow * (p.run + 1) <= kPackSlab;
  return p;
}

__global__ __launch_bounds__(RTG_THREADS) void pack_kernel(const RtgPackJob* jobs, int n_jobs, int lds_floats,
                                                           const float* params, const float* scales, float* packed) {
  // the grid is the concatenation of the jobs' block ranges (RtgPackJob.first_block / n_blocks): the job of this block is
  // the last one that starts at or before it (every thread tests one job; a 2-D grid of (blocks of the largest job) x
  // jobs spent most of the launch dispatching blocks that had nothing to do)
  int cnt = 0;
  for (int base = 0; base < n_jobs; base += RTG_THREADS) {
    const int i = base + (int)threadIdx.x;
    cnt += __syncthreads_count(i < n_jobs && jobs[i].first_block <= (int)blockIdx.x);
  }
  const RtgPackJob j = jobs[cnt - 1];
  const unsigned bid = blockIdx.x - (unsigned)j.first_block, nb = (unsigned)j.n_blocks;
  const int TM = j.tile_m, KK = 64 / TM, CPN = RTG_CK / KK;
  const int n_mt = (j.Mg + TM - 1) / TM, n_cc = (j.Cg + RTG_CK - 1) / RTG_CK;
  const unsigned n_e = (unsigned)j.dst_size;
  const int lane = threadIdx.x & 63;
  if (j.mode == RTG_PACK_GMFMA_FWD) {
    // [g][oc][ci][44]: row (g * Mg + oc) of v is (ci, tap)-major with 41 taps; taps 41 .. 43 of the image are zero
    const unsigned K = (unsigned)j.K, KP = (unsigned)j.S, Cg = (unsigned)j.Cg;      // (S carries the padded tap count)
    if (j.KH > 0) {
      // the pair image of the layer with 8 output channels per group (rtg_gmfma.hip, gmfma_pair_kernel): [g][16][ci][KP],
      // row (r, oc) holds w[g * 8 + oc][ci][u - KH * r] (KH = the layer's stride), zero outside the 41 taps
      const unsigned Mg = (unsigned)j.Mg;               // 8
      for (unsigned e = bid * RTG_THREADS + threadIdx.x; e < n_e; e += nb * RTG_THREADS) {
        const unsigned u = e % KP, rc = e / KP;         // rc = (g * 2 * Mg + r * Mg + oc) * Cg + ci
        const unsigned ci = rc % Cg, vrow = rc / Cg;
        const unsigned g = vrow / (2 * Mg), m = vrow % (2 * Mg), r = m / Mg, oc = m % Mg;
        const unsigned row = g * Mg + oc;
        const int t = (int)u - (int)r * j.KH;
        packed[j.dst_off + e] =
            (t >= 0 && t < (int)K) ? params[j.v_off + ((long long)row * Cg + ci) * K + t] * scales[j.scale_off + row] : 0.f;
      }
      return;
    }
    for (unsigned e = bid * RTG_THREADS + threadIdx.x; e < n_e; e += nb * RTG_THREADS) {
      const unsigned t = e % KP, rc = e / KP;           // rc = row * Cg + ci
      const unsigned row = rc / Cg;
      packed[j.dst_off + e] = t < K ? params[j.v_off + (long long)rc * K + t] * scales[j.scale_off + row] : 0.f;
    }
    return;
  }
  if (j.mode == RTG_PACK_GCONV_FWD || j.mode == RTG_PACK_GCONV_BWD) {
    // the vector-ALU kernels' plain orders (rtg_gconv.hip): w'[g][ci][t][oc] = v[g * Mg + oc][ci][t] * scale[g * Mg + oc]
    // (forward), w'[g][oc][t][ci] = the same element (backward-data); the source row of v is (ci, t)-major
    const unsigned CK = (unsigned)(j.Cg * j.K);
    for (unsigned e = bid * RTG_THREADS + threadIdx.x; e < n_e; e += nb * RTG_THREADS) {
      unsigned row, sin;
      if (j.mode == RTG_PACK_GCONV_FWD) {
        const unsigned oc = e % (unsigned)j.Mg, r = (e / (unsigned)j.Mg) % CK, g = e / ((unsigned)j.Mg * CK);
        row = g * (unsigned)j.Mg + oc;
        sin = r;
      } else {
        const unsigned ci = e % (unsigned)j.Cg, t = (e / (unsigned)j.Cg) % (unsigned)j.K;
        row = e / CK;
        sin = ci * (unsigned)j.K + t;
      }
      packed[j.dst_off + e] = params[j.v_off + (long long)row * CK + sin] * scales[j.scale_off + row];
    }
    return;
  }
  if (j.bf16 && j.frag16) {
    // the bf16 fragment image of rtg_dconv.hip: [16-row tile][32-channel chunk][tap][kgrp 4][row 16][8 bf16], channel
    // 8 * kgrp + i of the chunk in element i; one 32-bit slot = elements (2 * slot, 2 * slot + 1)
    const int n_c32 = (j.Cg + 31) / 32;
    for (unsigned e = bid * RTG_THREADS + threadIdx.x; e < n_e; e += nb * RTG_THREADS) {
      const int slot = (int)(e & 3u), m = (int)((e >> 2) & 15u), kg = (int)((e >> 6) & 3u);
      unsigned t2 = e >> 8;
      const int tap2 = (int)(t2 % j.K); t2 /= j.K;
      const int cc2 = (int)(t2 % n_c32);
      const int mt2 = (int)(t2 / n_c32);
      unsigned bits = 0;
      for (int h = 0; h < 2; ++h) {
        const float v = pack_logical(j, params, scales, 0, mt2 * 16 + m, cc2 * 32 + 8 * kg + 2 * slot + h, tap2);
        bits |= (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v) << (16 * h);
      }
      packed[j.dst_off + e] = __builtin_bit_cast(float, bits);
    }
    return;
  }
  if (j.bf16) {
    // two bf16 per 32-bit slot: elements (2e, 2e+1) of [g][mt][cc][tap][mfma][lane][4]
    for (unsigned e = bid * RTG_THREADS + threadIdx.x; e < n_e; e += nb * RTG_THREADS) {
      const int NMF = TM == 32 ? 2 : 1;
      unsigned bits = 0;
      for (int h = 0; h < 2; ++h) {
        unsigned t2 = 2 * e + h;
        const int el = (int)(t2 % 4); t2 /= 4;
        const int ln = (int)(t2 % 64); t2 /= 64;
        const int mf = (int)(t2 % NMF); t2 /= NMF;
        const int tap2 = (int)(t2 % j.K); t2 /= j.K;
        const int cc2 = (int)(t2 % n_cc); t2 /= n_cc;
        const int mt2 = (int)(t2 % n_mt); t2 /= n_mt;
        const int g2 = (int)t2;
        const int kk2 = ln / TM, m2 = ln - kk2 * TM;
        const int c2 = cc2 * RTG_CK + (TM == 32 ? 8 * mf + 4 * kk2 : 4 * kk2) + el;
        const float v = pack_logical(j, params, scales, g2, mt2 * TM + m2, c2, tap2);
        const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)v);
        bits |= (unsigned)b << (16 * h);
      }
      packed[j.dst_off + e] = __builtin_bit_cast(float, bits);
    }
    return;
  }
  // ---- fp32 images.  Every image is a sequence of slabs, one per (row tile, 16-channel chunk): K steps of 256 (frag16)
  // or 16 * tile_m floats, contiguous in the destination.  A slab's SOURCE elements are, per source row, one contiguous run
  // of the weight tensor ((channel, tap) pairs of the chunk: forward; (row, tap) pairs of the tile: stride-1
  // backward-data; whole taps of the tile's channels: polyphase), so the block copies those runs into LDS with coalesced
  // loads and emits the slab in destination order from there — a lane-per-destination gather touched 64 cache lines per
  // wave-load (the source stride between the rows of a tile is a whole weight row) and ran at 1.2 TB/s.  The tap-major
  // and 2-D backward-data images (short / strided source runs) keep the gather, one step per wave.
  const unsigned step_sz = j.frag16 ? 256u : (unsigned)(RTG_CK * TM);        // 256 or 512
  const int TG = (j.K + KK - 1) / KK;                                        // tap-major: tap groups per channel
  const int n_grp = (j.Cg * TG + CPN - 1) / CPN;
  const int n_mt16 = (j.Mg + 15) / 16;
  const float invS = 1.0f / (float)(j.S > 0 ? j.S : 1);
  const int inner = j.src_inner_c * j.src_K;
  const int n_co = j.KH > 0 ? j.Cg / j.KH : j.Cg;
  const float inv_nco = 1.0f / (float)(n_co > 0 ? n_co : 1);
  const int RT = j.frag16 ? 16 : TM;                                         // rows of a row tile
  const int n_rt = j.frag16 ? n_mt16 : n_mt;
  // (sized by the launch to the largest slab of its jobs — rtg_pack_job_lds —: 10 KB for the 5-tap layers instead of the
  // 31 KB of the largest slab the staging serves, three times the workgroups in flight per CU)
  extern __shared__ float slab[];
  const PackGeom pg = pack_geom(j);
  if (pg.staged && pg.nrow * (pg.run + 1) <= lds_floats) {
    // run length per source row and rows per slab: forward: RT rows x 16 K; stride-1 backward-data: 16 rows x RT K;
    // polyphase: 16 rows x (channels the tile's rows cover) x src_K
    const int run = pg.run, nrow = pg.nrow, pitch = run + 1;
    {
      const unsigned n_slab = (unsigned)j.groups * n_rt * n_cc;
      for (unsigned sl = bid; sl < n_slab; sl += nb) {
        unsigned t = sl;
        const int cc = (int)(t % n_cc); t /= n_cc;
        const int mt = (int)(t % n_rt); t /= n_rt;
        const int g = (int)t;
        const int m0 = mt * RT, c0 = cc * RTG_CK;
        const int ch0 = m0 / (j.S > 0 ? j.S : 1);                            // polyphase: first channel of the tile's rows
        // ---- source runs -> LDS (scaled by the row's g / ||v||), zero where the row / channel does not exist
        for (int f = threadIdx.x; f < nrow * run; f += RTG_THREADS) {
          const int r = f / run, u = f - r * run;
          long long srow;
          int sin;
          bool ok;
          if (j.mode == RTG_PACK_FWD) {
            ok = m0 + r < j.Mg && c0 * j.src_K + u < j.Cg * j.src_K;
            srow = (long long)g * j.Mg + m0 + r;
            sin = c0 * j.src_K + u;
          } else if (j.mode == RTG_PACK_DGRAD_S1) {
            ok = c0 + r < j.Cg && m0 * j.src_K + u < j.Mg * j.src_K;
            srow = (long long)g * j.Cg + c0 + r;
            sin = m0 * j.src_K + u;
          } else {
            ok = c0 + r < j.Cg && ch0 * j.src_K + u < inner;
            srow = j.mode == RTG_PACK_DGRAD_POLY ? (long long)g * j.Cg + c0 + r : (long long)(c0 + r);
            sin = ch0 * j.src_K + u;
          }
          slab[r * pitch + u] = ok ? params[j.v_off + srow * inner + sin] * scales[j.scale_off + srow] : 0.f;
        }
        __syncthreads();
        // ---- the slab in destination order
        float* dst = packed + j.dst_off + (size_t)sl * j.K * step_sz;
        const int step_shift = step_sz == 512u ? 9 : 8;
        for (unsigned e = threadIdx.x; e < (unsigned)j.K * step_sz; e += RTG_THREADS) {
          const int tap = (int)(e >> step_shift);
          const unsigned in = e & (step_sz - 1);
          int ml, cl;
          if (j.frag16) { ml = (in >> 2) & 15; cl = 4 * (in & 3) + (in >> 6); }
          else { ml = in & (TM - 1); cl = (in / 64) * KK + ((in / TM) & (KK - 1)); }
          float val = 0.f;
          if (m0 + ml < j.Mg && c0 + cl < j.Cg) {
            if (j.mode == RTG_PACK_FWD) {
              val = slab[ml * pitch + cl * j.src_K + tap];
            } else if (j.mode == RTG_PACK_DGRAD_S1) {
              val = slab[cl * pitch + ml * j.src_K + (j.src_K - 1 - tap)];
            } else {
              const int m = m0 + ml;
              int ch = (int)((float)m * invS);
              int r = m - ch * j.S;
              if (r < 0) { --ch; r += j.S; }
              else if (r >= j.S) { ++ch; r -= j.S; }
              const int jj = r + (j.K - 1 - tap) * j.S;
              if (jj < j.src_K) val = slab[cl * pitch + (ch - ch0) * j.src_K + jj];
            }
          }
          dst[e] = val;
        }
        __syncthreads();
      }
      return;
    }
  }
  // ---- gather path: a wave packs one whole step per iteration (the step's coordinates once per 256 / 512 elements)
  const unsigned n_steps = n_e / step_sz;                                    // (dst_size is a multiple of the step size)
  const unsigned wave0 = (bid * RTG_THREADS + threadIdx.x) >> 6, wstride = (nb * RTG_THREADS) >> 6;
  for (unsigned st = wave0; st < n_steps; st += wstride) {
    unsigned t = __builtin_amdgcn_readfirstlane(st);
    // ---- the step's coordinates, once per wave
    int tap = 0, cc = 0, mt, g = 0, grp = 0;
    if (j.frag16) {
      tap = (int)(t % j.K); t /= j.K;
      cc = (int)(t % n_cc); t /= n_cc;
      mt = (int)(t % n_mt16);
    } else if (j.tap_major) {
      grp = (int)(t % n_grp); t /= n_grp;
      mt = (int)(t % n_mt); t /= n_mt;
      g = (int)t;
    } else {
      tap = (int)(t % j.K); t /= j.K;
      cc = (int)(t % n_cc); t /= n_cc;
      mt = (int)(t % n_mt); t /= n_mt;
      g = (int)t;
    }
    float* dst = packed + j.dst_off + (size_t)st * step_sz;
    for (unsigned in = lane; in < step_sz; in += 64) {
      // ---- the element's (row, channel, tap) within the step: shifts only
      int m, c, tp = tap;
      if (j.frag16) {
        // [kgrp 4][m 16][kq 4], channel = 4 * kq + kgrp of the chunk (rtg_dconv.hip)
        m = mt * 16 + ((in >> 2) & 15);
        c = cc * RTG_CK + 4 * (in & 3) + (in >> 6);
      } else {
        const int mm = in & (TM - 1), kk = (in / TM) & (KK - 1), cp = in / 64;
        m = mt * TM + mm;
        if (j.tap_major) {
          // k-step = group * CPN + cp = (channel, tap group); tap = tap group * KK + kk
          const int ks = grp * CPN + cp;
          c = ks / TG;
          tp = (ks - c * TG) * KK + kk;
          if (tp >= j.K) c = j.Cg;             // padding taps: zero
        } else {
          c = cc * RTG_CK + cp * KK + kk;
        }
      }
      // ---- the source element (pack_logical in 32-bit arithmetic; divisions by the stride / the kernel rows through the
      // float reciprocal, exact for operands below 2^24 after one correction step)
      float val = 0.f;
      if (m < j.Mg && c < j.Cg) {
        int srow = -1, sin = 0;
        if (j.mode == RTG_PACK_FWD) {
          srow = g * j.Mg + m;
          sin = c * j.src_K + tp;
          if (j.kh_major) {
            int kh = (int)((float)c * inv_nco);
            int ci = c - kh * n_co;
            if (ci < 0) { --kh; ci += n_co; }
            else if (ci >= n_co) { ++kh; ci -= n_co; }
            sin = (ci * j.KH + kh) * j.src_K + tp;
          }
        } else if (j.mode == RTG_PACK_DGRAD_S1) {
          srow = g * j.Cg + c;
          sin = m * j.src_K + (j.src_K - 1 - tp);
        } else {
          int ch = (int)((float)m * invS);
          int r = m - ch * j.S;
          if (r < 0) { --ch; r += j.S; }
          else if (r >= j.S) { ++ch; r -= j.S; }
          const int jj = r + (j.K - 1 - tp) * j.S;
          if (jj < j.src_K) {
            if (j.mode == RTG_PACK_DGRAD_2D) {
              int kh = (int)((float)c * inv_nco);
              int co = c - kh * n_co;
              if (co < 0) { --kh; co += n_co; }
              else if (co >= n_co) { ++kh; co -= n_co; }
              srow = co;
              sin = (ch * j.KH + kh) * j.src_K + jj;
            } else if (j.mode == RTG_PACK_DGRAD_POLY) {
              srow = g * j.Cg + c;
              sin = ch * j.src_K + jj;
            } else {                           // RTG_PACK_CONVT_POLY (groups == 1): source [C_in][C_out][K]
              srow = c;
              sin = ch * j.src_K + jj;
            }
          }
        }
        if (srow >= 0) val = params[j.v_off + (long long)srow * inner + pack_src_index(j, sin)] * scales[j.scale_off + srow];
      }
      dst[in] = val;
    }
  }
}

// One block per (layer, row).  Every thread owns the elements tid, tid + 256, ... of the row and sums their split
// partials itself, splits in ascending order (bitwise reproducible, no cross-wave combine), 32 independent loads in
// flight per thread: the kernel is a pure stream over the partial buffers (0.3 GB per model and pass) and lives on
// memory-level parallelism.  The row sum is parked in LDS for the second pass (the weight-norm chain rule needs
// <dW_row, v_row> before any element of dv can be written).
__global__ __launch_bounds__(RTG_THREADS) void wn_bwd_kernel(const RtgWnBwdJob* jobs, const float* params,
                                                              const float* scales, const float* partials,
                                                              float* grads) {
  extern __shared__ float dw[];            // [inner] the reduced row
  __shared__ float red[4];
  const RtgWnBwdJob j = jobs[blockIdx.y];
  // element i of the tensor's row [ci][taps][rows] -> its column in the partials [ci][rows][taps] (t_rows > 0: a layer packed
  // with RtgPackJob.src_T), else i itself
  auto pcol = [&](int i) __attribute__((always_inline)) {
    if (j.t_rows <= 0) return i;
    const int ct = i / j.t_rows, rr = i - ct * j.t_rows;
    const int ci = ct / j.t_taps, t = ct - ci * j.t_taps;
    return (ci * j.t_rows + rr) * j.t_taps + t;
  };
  for (int r = blockIdx.x; r < j.rows; r += gridDim.x) {
    const float* v = params + j.v_off + (size_t)r * j.inner;
    const float* p0 = partials + j.part_off + (size_t)r * j.inner;
    float dot = 0.f;
    // 32 loads in flight per thread: two elements x 16 splits, or (rows no longer than the block) one element x 32
    // splits — the dependent chain of a row is splits / 32 memory latencies, whatever the split count (the narrow
    // generator layers have 256 splits, the big discriminator layers 6 .. 11).  A short last batch re-reads the last
    // split (cached) and drops the value.
    // bias column (one partial per split): requested first, 32 splits in flight, summed in ascending order by thread 0
    float bsum = 0.f;
    if (threadIdx.x == 0 && j.b_off >= 0) {
      const float* pb = partials + j.part_off + (size_t)j.rows * j.inner + r;
      for (int sp = 0; sp < j.splits; sp += 32) {
        float tb[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) tb[u] = pb[(size_t)(sp + u < j.splits ? sp + u : j.splits - 1) * j.part_stride];
#pragma unroll
        for (int u = 0; u < 32; ++u) bsum += (sp + u < j.splits) ? tb[u] : 0.f;
      }
    }
    // (rows of fewer than 1024 elements keep one element per thread: more splits in flight per row)
    const bool vec = j.inner >= 4 * RTG_THREADS && (j.inner & 3) == 0 && (j.part_off & 3) == 0 && (j.part_stride & 3) == 0 &&
                     j.t_rows <= 0;
    if (vec) {
      // 16-byte loads of the partials (the bulk of the traffic: splits x the row), four consecutive elements per thread,
      // 8 splits in flight; every element still sums its splits in ascending order (the same bits as the scalar path)
      const int n4 = j.inner >> 2;
      for (int i4 = threadIdx.x; i4 < n4; i4 += RTG_THREADS) {
        const f32x4* pa = reinterpret_cast<const f32x4*>(p0) + i4;
        const size_t st4 = (size_t)(j.part_stride >> 2);
        f32x4 sa = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < j.splits; sp += 8) {
          f32x4 ta[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) ta[u] = pa[(size_t)(sp + u < j.splits ? sp + u : j.splits - 1) * st4];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (sp + u < j.splits) { sa.x += ta[u].x; sa.y += ta[u].y; sa.z += ta[u].z; sa.w += ta[u].w; }
          }
        }
        const int i = 4 * i4;
        dw[i] = sa.x; dw[i + 1] = sa.y; dw[i + 2] = sa.z; dw[i + 3] = sa.w;
        // (the scalar path adds this thread's elements i, i + 256, ... in that order; here the thread owns 4i4 .. 4i4+3:
        // the block-wide dot product is summed in another order — rounding-level difference in d g and the k2 term)
        dot += sa.x * v[i];
        dot += sa.y * v[i + 1];
        dot += sa.z * v[i + 2];
        dot += sa.w * v[i + 3];
      }
    } else if (j.inner <= RTG_THREADS) {
      const int i = threadIdx.x;
      if (i < j.inner) {
        const float* pa = p0 + pcol(i);
        float sa = 0.f;
        for (int sp = 0; sp < j.splits; sp += 32) {
          float ta[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) {
            const int q = sp + u < j.splits ? sp + u : j.splits - 1;
            ta[u] = pa[(size_t)q * j.part_stride];
          }
#pragma unroll
          for (int u = 0; u < 32; ++u) sa += (sp + u < j.splits) ? ta[u] : 0.f;
        }
        dw[i] = sa;
        dot += sa * v[i];
      }
    } else {
      for (int i = threadIdx.x; i < j.inner; i += 2 * RTG_THREADS) {
        const bool two = i + RTG_THREADS < j.inner;
        const float* pa = p0 + pcol(i);
        const float* pb = p0 + pcol(two ? i + RTG_THREADS : i);
        float sa = 0.f, sb = 0.f;
        for (int sp = 0; sp < j.splits; sp += 16) {
          float ta[16], tb[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int q = sp + u < j.splits ? sp + u : j.splits - 1;
            ta[u] = pa[(size_t)q * j.part_stride];
            tb[u] = pb[(size_t)q * j.part_stride];
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const bool ok = sp + u < j.splits;
            sa += ok ? ta[u] : 0.f;
            sb += ok ? tb[u] : 0.f;
          }
        }
        dw[i] = sa;
        dot += sa * v[i];
        if (two) {
          dw[i + RTG_THREADS] = sb;
          dot += sb * v[i + RTG_THREADS];
        }
      }
    }
    dot = rtg_block_sum(dot, red);
    const float scale = scales[j.scale_off + r], inv_n = scales[j.scale_off + j.rows + r];
    const float k2 = scale * dot * inv_n * inv_n;
    float* dv = grads + j.v_off + (size_t)r * j.inner;
    for (int i = threadIdx.x; i < j.inner; i += RTG_THREADS) dv[i] += scale * dw[i] - k2 * v[i];   // own elements only
    if (threadIdx.x == 0) {
      grads[j.g_off + r] += dot * inv_n;
      if (j.b_off >= 0) grads[j.b_off + r] += bsum;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int rtg_weightnorm_scales(const RtgNormJob* jobs_dev, int n_jobs, int max_rows, const float* params,
                                     float* scales, void* stream) {
  if (!jobs_dev || !params || !scales) return RTG_ENULL;
  if (n_jobs < 1 || n_jobs > 65535 || max_rows < 1) return RTG_EINVAL;
  dim3 grid(max_rows > 1024 ? 1024 : max_rows, n_jobs);
  RTG_KLAUNCH(wn_scales_kernel, grid, dim3(RTG_THREADS), 0, (hipStream_t)stream, jobs_dev, params, scales);
  return rtg_launch_status();
}

extern "C" int rtg_pack_job_blocks(const RtgPackJob* job) {
  if (!job || job->dst_size < 1 || (job->tile_m != 16 && job->tile_m != 32)) return -1;
  if (job->dst_size >= (1ll << 30)) return -1;                              // 32-bit index decode in the kernel (2e + 1 must fit)
  const PackGeom pg = pack_geom(*job);
  long long n;
  if (pg.staged) {
    n = (long long)job->groups * pg.n_rt * pg.n_cc;                          // one slab per block
  } else if (job->bf16 || job->mode >= RTG_PACK_GCONV_FWD) {
    n = (job->dst_size + RTG_THREADS * 4 - 1) / (RTG_THREADS * 4);
  } else {
    const long long n_steps = job->dst_size / (job->frag16 ? 256 : RTG_CK * job->tile_m);
    n = (n_steps + 7) / 8;                                                   // two steps per wave
  }
  return (int)(n < 1 ? 1 : (n > 4096 ? 4096 : n));
}

// floats of LDS job *job (HOST memory) stages a slab in (0: it takes the gather path)
extern "C" int rtg_pack_job_lds(const RtgPackJob* job) {
  if (!job || (job->tile_m != 16 && job->tile_m != 32)) return -1;
  const PackGeom pg = pack_geom(*job);
  return pg.staged ? pg.nrow * (pg.run + 1) : 0;
}

extern "C" int rtg_weights_pack(const RtgPackJob* jobs_dev, int n_jobs, long long total_blocks, int lds_floats,
                                const float* params, const float* scales, float* packed, void* stream) {
  if (!jobs_dev || !params || !scales || !packed) return RTG_ENULL;
  if (n_jobs < 1 || n_jobs > 65535 || total_blocks < 1 || lds_floats < 0 || lds_floats > kPackSlab) return RTG_EINVAL;
  if (total_blocks >= (1ll << 31)) return RTG_ERANGE;
  RTG_KLAUNCH(pack_kernel, dim3((unsigned)total_blocks), dim3(RTG_THREADS), (size_t)lds_floats * sizeof(float),
              (hipStream_t)stream, jobs_dev, n_jobs, lds_floats, params, scales, packed);
  return rtg_launch_status();
}

extern "C" int rtg_weightnorm_backward(const RtgWnBwdJob* jobs_dev, int n_jobs, int max_rows, int max_inner,
                                       const float* params, const float* scales, const float* partials, float* grads,
                                       void* stream) {
  if (!jobs_dev || !params || !scales || !partials || !grads) return RTG_ENULL;
  if (n_jobs < 1 || n_jobs > 65535 || max_rows < 1 || max_inner < 1 || max_inner > 8192) return RTG_EINVAL;
  dim3 grid(max_rows > 1024 ? 1024 : max_rows, n_jobs);
  RTG_KLAUNCH(wn_bwd_kernel, grid, dim3(RTG_THREADS), (size_t)max_inner * sizeof(float),
                     (hipStream_t)stream,
                     jobs_dev, params, scales, partials, grads);
  return rtg_launch_status();
}
