// halo.hip -- HBM-bound data-movement kernels of the forecast step, gfx950.
//
//   pad2d_fwd      PeriodicPadding2D.call (DLWP/custom.py:191-214), FillPadding2D.call (custom.py:359-402), keras
//                  ZeroPadding2D -- one pass instead of the reference's 2-3 concat copies per layer
//   pad2d_bwd      adjoint (halo fold-back), gather form, deterministic
//   maxpool2 / upsample2 fwd+bwd   keras MaxPooling2D(2) / UpSampling2D(2)  (examples/train.py:171,181,191,201)
//   copy_channels  slice_layer (custom.py:675-692) / keras concatenate(axis=1)
//   series_merge_time  the final reshape/transpose of predict_timeseries (DLWP/model/models.py:294-300)
//
// All of these move each byte once: algorithmic bytes = bytes(in) + bytes(out); roofline = HBM.
#include "common.h"
#include "tape.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static inline int grid_for(long long work_items, int block, int cu_count) {
  long long want = (work_items + block - 1) / block;
  long long cap = (long long)cu_count * 8;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

// ------------------------------------------------------------------------------------------------------------------ //
// pad2d forward: LDS-staged rows.
// View: [outer, H, RI] -> [outer, Ho, RO] with RI = W*inner, RO = Wo*inner floats per row.  One wave owns one output
// row at a time: it pulls the source row into its LDS slice with aligned 16-byte loads (coalesced, each HBM byte read
// once), then emits the output row with aligned 16-byte stores whose lanes pick their 4 values from LDS at the
// shifted / wrapped / clamped position.  The shift by `left*inner` elements that makes direct vector copies
// misaligned is absorbed by LDS.
// ------------------------------------------------------------------------------------------------------------------ //
// column map for padding amounts <= W (validated by the host): one conditional add instead of an integer modulo
__device__ __forceinline__ int pad_map_col(int p, int n, int mode) {
  if (p >= 0 && p < n) return p;
  if (mode == DLWP_PAD_ZERO) return -1;
  if (mode == DLWP_PAD_EDGE) return p < 0 ? 0 : n - 1;
  if (mode == DLWP_PAD_REFLECT) return p < 0 ? -p : 2 * n - 2 - p;
  if (mode == DLWP_PAD_SYMMETRIC) return p < 0 ? -p - 1 : 2 * n - 1 - p;
  return p < 0 ? p + n : p - n;
}

template <int VEC, bool INNER1>
__global__ __launch_bounds__(256) void pad2d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int outer,
                                                        int H, int W, int inner, int Ho, int Wo, int top, int left,
                                                        int mode_h, int mode_w, int row_lds /* floats per row slot */) {
  // Every wave works alone on its own LDS slots (ROWS output rows per trip): no workgroup barrier anywhere -- a
  // wave's DS instructions execute in program order, so its ds_reads see its own preceding ds_writes.  The ROWS rows
  // are walked as ONE flat list of VEC-wide slots so that all 64 lanes stay busy although a row is only ~46 slots.
  constexpr int ROWS = 4;
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* slot = lds + wave * ROWS * row_lds;
  const int RI = W * inner, RO = Wo * inner;
  const int RIV = RI / VEC, ROV = RO / VEC;
  const long long n_rows = (long long)outer * Ho;
  const long long stride = (long long)gridDim.x * 4 * ROWS;
  for (long long r0 = ((long long)blockIdx.x * 4 + wave) * ROWS; r0 < n_rows; r0 += stride) {
    // source row (in rows of the whole input tensor) of each of the ROWS output rows, -1 = zero row / past the end
    long long srow[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      const long long r = r0 + k;
      srow[k] = -1;
      if (r < n_rows) {
        const int o = (int)(r / Ho);
        const int hs = dlwp_map_coord((int)(r - (long long)o * Ho) - top, H, mode_h);
        if (hs >= 0) srow[k] = (long long)o * H + hs;
      }
    }
    for (int j = lane; j < ROWS * RIV; j += 64) {
      const int k = (j >= RIV) + (j >= 2 * RIV) + (j >= 3 * RIV);
      const int c = (j - k * RIV) * VEC;
      const long long sr = k == 0 ? srow[0] : (k == 1 ? srow[1] : (k == 2 ? srow[2] : srow[3]));
      if (sr >= 0) *(vec_t*)(slot + k * row_lds + c) = *(const vec_t*)(x + sr * RI + c);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = lane; j < ROWS * ROV; j += 64) {
      const int k = (j >= ROV) + (j >= 2 * ROV) + (j >= 3 * ROV);
      const int c = (j - k * ROV) * VEC;
      const long long r = r0 + k;
      if (r >= n_rows) continue;
      const long long sr = k == 0 ? srow[0] : (k == 1 ? srow[1] : (k == 2 ? srow[2] : srow[3]));
      const float* row = slot + k * row_lds;
      vec_t v;
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        const int e = c + q;
        int wo = e, ch = 0;
        if (!INNER1) {
          wo = e / inner;
          ch = e - wo * inner;
        }
        const int ws = pad_map_col(wo - left, W, mode_w);
        v[q] = (sr >= 0 && ws >= 0) ? row[(ws >= 0 ? ws : 0) * (INNER1 ? 1 : inner) + ch] : 0.f;
      }
      *(vec_t*)(y + r * RO + c) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// pad2d backward (adjoint), row-staged like the forward: a wave owns ROWS dx rows at a time, pulls their INTERIOR dy rows
// (row h + top of the padded gradient: where all but 2 * halo / H of the rows get everything from) into its LDS slice with
// aligned vector loads -- every dy byte of those rows read once, coalesced -- and emits the dx rows with aligned vector
// stores whose lanes add the column images (interior column + wrapped / clamped halo columns) out of LDS.  The few dx rows
// that are also the image of halo ROWS (the first / last `bottom` / `top` rows of a periodic axis, row 0 / H-1 of an edge
// axis) add those rows straight from global memory.  Fixed summation order per element: deterministic.
template <int VEC, bool INNER1>
__global__ __launch_bounds__(256) void pad2d_bwd_rows_kernel(const float* __restrict__ dy, float* __restrict__ dx, int outer,
                                                             int H, int W, int inner, int Ho, int Wo, int top, int bottom,
                                                             int left, int right, int mode_h, int mode_w, int row_lds) {
  constexpr int ROWS = 4;
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* slot = lds + wave * ROWS * row_lds;
  const int RI = W * inner, RO = Wo * inner;
  const int RIV = RI / VEC, ROV = RO / VEC;
  const long long n_rows = (long long)outer * H;
  const long long stride = (long long)gridDim.x * 4 * ROWS;
  // sum of the dy columns of one dy row (rp: LDS or global) that are images of dx column w, channel ch
  auto col_sum = [&](const float* rp, int w, int ch) {
    const int in = INNER1 ? 1 : inner;
    float s = rp[(w + left) * in + ch];
    if (mode_w == DLWP_PAD_WRAP) {
      if (w >= W - left) s += rp[(w - (W - left)) * in + ch];
      if (w < right) s += rp[(left + W + w) * in + ch];
    } else if (mode_w == DLWP_PAD_EDGE) {
      if (w == 0)
        for (int c = 0; c < left; ++c) s += rp[c * in + ch];
      if (w == W - 1)
        for (int c = left + W; c < Wo; ++c) s += rp[c * in + ch];
    } else if (mode_w >= DLWP_PAD_REFLECT) {   // mirror halos: every halo column is the image of exactly one column
      for (int c = 0; c < left; ++c)
        if (pad_map_col(c - left, W, mode_w) == w) s += rp[c * in + ch];
      for (int c = left + W; c < Wo; ++c)
        if (pad_map_col(c - left, W, mode_w) == w) s += rp[c * in + ch];
    }
    return s;
  };
  for (long long r0 = ((long long)blockIdx.x * 4 + wave) * ROWS; r0 < n_rows; r0 += stride) {
    for (int j = lane; j < ROWS * ROV; j += 64) {
      const int k = (j >= ROV) + (j >= 2 * ROV) + (j >= 3 * ROV);
      const int c = (j - k * ROV) * VEC;
      const long long r = r0 + k;
      if (r < n_rows) {
        const long long o = r / H;
        const int hh = (int)(r - o * H);
        *(vec_t*)(slot + k * row_lds + c) = *(const vec_t*)(dy + (o * Ho + hh + top) * RO + c);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = lane; j < ROWS * RIV; j += 64) {
      const int k = (j >= RIV) + (j >= 2 * RIV) + (j >= 3 * RIV);
      const int c = (j - k * RIV) * VEC;
      const long long r = r0 + k;
      if (r >= n_rows) continue;
      const long long o = r / H;
      const int hh = (int)(r - o * H);
      const float* row = slot + k * row_lds;
      const float* img = dy + o * Ho * RO;        // this image's padded gradient, for the halo-row images
      vec_t v;
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        const int e = c + q;
        int w = e, ch = 0;
        if (!INNER1) {
          w = e / inner;
          ch = e - w * inner;
        }
        float acc = col_sum(row, w, ch);
        if (mode_h == DLWP_PAD_WRAP) {
          if (hh >= H - top) acc += col_sum(img + (long long)(hh - (H - top)) * RO, w, ch);
          if (hh < bottom) acc += col_sum(img + (long long)(top + H + hh) * RO, w, ch);
        } else if (mode_h == DLWP_PAD_EDGE) {
          if (hh == 0)
            for (int rr = 0; rr < top; ++rr) acc += col_sum(img + (long long)rr * RO, w, ch);
          if (hh == H - 1)
            for (int rr = top + H; rr < Ho; ++rr) acc += col_sum(img + (long long)rr * RO, w, ch);
        } else if (mode_h >= DLWP_PAD_REFLECT) {
          for (int rr = 0; rr < top; ++rr)
            if (pad_map_col(rr - top, H, mode_h) == hh) acc += col_sum(img + (long long)rr * RO, w, ch);
          for (int rr = top + H; rr < Ho; ++rr)
            if (pad_map_col(rr - top, H, mode_h) == hh) acc += col_sum(img + (long long)rr * RO, w, ch);
        }
        v[q] = acc;
      }
      *(vec_t*)(dx + r * RI + c) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// r5: the FLAT row-group kernels -- NCHW (inner = 1), 16-byte aligned tensors, ANY row length.
// r4's row-staged kernels moved rows as 16-, 8- or 4-byte units according to the row length, four rows per wave: 92-float rows
// (44 x 90 maps) went as 8-byte units at 0.40 of the HBM peak, 47-float rows (22 x 45) as single floats at 0.21, and even
// whole-unit rows had only ~2.9 KB (180 floats) or ~0.75 KB (47 floats) of loads in flight per wave.  Here a wave owns a GROUP
// of ROWS consecutive output rows, ROWS in {4, 8, 16} chosen so that a group is >= ~2.5 KB: a multiple of 4 rows starts at a
// multiple of 4 floats whatever the row length, so the group is STORED as aligned 16-byte vectors of the flat output (a vector
// may span two rows: every element finds its own row and column through a multiply-high division); a source row that starts
// misaligned is pulled into LDS through the aligned 16-byte window around it (<= 3 floats of its neighbours ride along, the
// low two bits of its base say where it starts inside its slot).  Per-row bases live in a small LDS table of the wave.
// ------------------------------------------------------------------------------------------------------------------ //
__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned magic) {   // n / d for n * d < 2^32; magic 0: d = 1
  return magic ? __umulhi(n, magic) : n;
}

template <int ROWS, int MH = -1, int MW = -1>
__global__ __launch_bounds__(256) void pad2d_fwd_flat_kernel(const float* __restrict__ x, float* __restrict__ y, int outer, int H,
                                                             int W, int Ho, int Wo, int top, int left, int mode_h_rt, int mode_w_rt,
                                                             int row_lds, unsigned magic_nv, unsigned magic_ro) {
  const int mode_h = MH >= 0 ? MH : mode_h_rt, mode_w = MW >= 0 ? MW : mode_w_rt;   // (see pad2d_bwd_flat_kernel)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* slot = lds + wave * ROWS * row_lds;
  long long* tbl = (long long*)(lds + 4 * ROWS * row_lds) + wave * ROWS;      // first source element of each row, -1: zero row
  const int RI = W, RO = Wo;
  const int NV = (RI + 6) >> 2;                        // 16-byte units of a row's aligned window, at most
  const long long n_rows = (long long)outer * Ho;
  const long long total_in = (long long)outer * H * RI;
  const long long stride = (long long)gridDim.x * 4 * ROWS;
  for (long long r0 = ((long long)blockIdx.x * 4 + wave) * ROWS; r0 < n_rows; r0 += stride) {
    if (lane < ROWS) {
      const long long r = r0 + lane;
      long long b = -1;
      if (r < n_rows) {
        const long long o = r / Ho;
        const int hs = dlwp_map_coord((int)(r - o * Ho) - top, H, mode_h);
        if (hs >= 0) b = (o * H + hs) * RI;
      }
      tbl[lane] = b;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = lane; j < ROWS * NV; j += 64) {
      const int k = (int)udiv_magic((unsigned)j, magic_nv);
      const int i = j - k * NV;
      const long long b = tbl[k];
      if (b < 0) continue;
      const int off = (int)(b & 3);
      if (4 * i >= off + RI) continue;
      const long long a = (b & ~3ll) + 4 * i;
      float* d = slot + k * row_lds + 4 * i;
      if (a + 4 <= total_in) {
        *(f32x4*)d = __builtin_nontemporal_load((const f32x4*)(x + a));
      } else {                                           // the tensor's last, partial unit
        for (int q = 0; q < 4; ++q)
          if (a + q < total_in) d[q] = x[a + q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int rows_here = (int)(n_rows - r0 < ROWS ? n_rows - r0 : ROWS);
    const int valid = rows_here * RO;                    // floats of this group that exist
    for (int j = lane; 4 * j < valid; j += 64) {
      const int k0 = (int)udiv_magic((unsigned)(4 * j), magic_ro);
      const long long b0 = tbl[k0], b1 = tbl[k0 + 1 < ROWS ? k0 + 1 : k0];
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = 4 * j + q;
        int k = k0;
        long long b = b0;
        if (e >= (k0 + 1) * RO) {                          // (RO < 4: a vector may span several rows -- walk on)
          k = k0 + 1;
          b = b1;
          while (e >= (k + 1) * RO) {
            ++k;
            b = k < ROWS ? tbl[k] : -1;
          }
        }
        const int wo = e - k * RO;
        const int ws = pad_map_col(wo - left, W, mode_w);
        v[q] = (b >= 0 && ws >= 0 && e < valid) ? slot[k * row_lds + (int)(b & 3) + (ws >= 0 ? ws : 0)] : 0.f;
      }
      float* dst = y + r0 * RO + 4 * j;
      if (4 * j + 4 <= valid) {
        __builtin_nontemporal_store(v, (f32x4*)dst);
      } else {
        for (int q = 0; q < 4; ++q)
          if (4 * j + q < valid) dst[q] = v[q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// backward: a wave owns ROWS dx rows, stages their INTERIOR dy rows, adds the column images out of LDS and the halo-row images
// straight from global memory (as pad2d_bwd_rows_kernel), stores the group as flat 16-byte vectors.  Deterministic.
// MH / MW >= 0: the halo modes as compile-time constants -- the composite halo of every network here (zero rows, periodic
// columns) gets an instance of a few hundred instructions; with run-time modes the four elements of a vector each carry the whole
// mode switch (10 k lines of code: the r2-r4 kernel ran at 0.3-0.48 of the HBM peak out of the instruction cache).
template <int ROWS, int MH = -1, int MW = -1>
__global__ __launch_bounds__(256) void pad2d_bwd_flat_kernel(const float* __restrict__ dy, float* __restrict__ dx, int outer, int H,
                                                             int W, int Ho, int Wo, int top, int bottom, int left, int right,
                                                             int mode_h_rt, int mode_w_rt, int row_lds, unsigned magic_nv,
                                                             unsigned magic_ri) {
  const int mode_h = MH >= 0 ? MH : mode_h_rt, mode_w = MW >= 0 ? MW : mode_w_rt;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* slot = lds + wave * ROWS * row_lds;
  long long* tbl = (long long*)(lds + 4 * ROWS * row_lds) + wave * ROWS;      // first dy element of each dx row's interior row
  int* hht = (int*)((long long*)(lds + 4 * ROWS * row_lds) + 4 * ROWS) + wave * ROWS;   // ... and the row's index in its image
  const int RI = W, RO = Wo;
  const int NV = (RO + 6) >> 2;
  const long long n_rows = (long long)outer * H;
  const long long total_dy = (long long)outer * Ho * RO;
  const long long stride = (long long)gridDim.x * 4 * ROWS;
  auto col_sum = [&](const float* rp, int w) {
    float s = rp[w + left];
    if (mode_w == DLWP_PAD_WRAP) {
      if (w >= W - left) s += rp[w - (W - left)];
      if (w < right) s += rp[left + W + w];
    } else if (mode_w == DLWP_PAD_EDGE) {
      if (w == 0)
        for (int c = 0; c < left; ++c) s += rp[c];
      if (w == W - 1)
        for (int c = left + W; c < Wo; ++c) s += rp[c];
    } else if (mode_w >= DLWP_PAD_REFLECT) {   // mirror halos: every halo column is the image of exactly one column
      for (int c = 0; c < left; ++c)
        if (pad_map_col(c - left, W, mode_w) == w) s += rp[c];
      for (int c = left + W; c < Wo; ++c)
        if (pad_map_col(c - left, W, mode_w) == w) s += rp[c];
    }
    return s;
  };
  for (long long r0 = ((long long)blockIdx.x * 4 + wave) * ROWS; r0 < n_rows; r0 += stride) {
    if (lane < ROWS) {
      const long long r = r0 + lane;
      long long b = -1;
      int hh = 0;
      if (r < n_rows) {
        const long long o = r / H;
        hh = (int)(r - o * H);
        b = (o * Ho + hh + top) * RO;
      }
      tbl[lane] = b;
      hht[lane] = hh;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = lane; j < ROWS * NV; j += 64) {
      const int k = (int)udiv_magic((unsigned)j, magic_nv);
      const int i = j - k * NV;
      const long long b = tbl[k];
      if (b < 0) continue;
      const int off = (int)(b & 3);
      if (4 * i >= off + RO) continue;
      const long long a = (b & ~3ll) + 4 * i;
      float* d = slot + k * row_lds + 4 * i;
      if (a + 4 <= total_dy) {
        *(f32x4*)d = __builtin_nontemporal_load((const f32x4*)(dy + a));
      } else {
        for (int q = 0; q < 4; ++q)
          if (a + q < total_dy) d[q] = dy[a + q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int rows_here = (int)(n_rows - r0 < ROWS ? n_rows - r0 : ROWS);
    const int valid = rows_here * RI;
    for (int j = lane; 4 * j < valid; j += 64) {
      const int k0 = (int)udiv_magic((unsigned)(4 * j), magic_ri);
      const long long b0 = tbl[k0], b1 = tbl[k0 + 1 < ROWS ? k0 + 1 : k0];
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = 4 * j + q;
        int k = k0;
        long long b = b0;
        if (e >= (k0 + 1) * RI) {                          // (RI < 4: a vector may span several rows -- walk on)
          k = k0 + 1;
          b = b1;
          while (e >= (k + 1) * RI) {
            ++k;
            b = k < ROWS ? tbl[k] : -1;
          }
        }
        const int w = e - k * RI;
        float acc = 0.f;
        if (b >= 0 && e < valid) {
          const int hh = hht[k];
          const float* row = slot + k * row_lds + (int)(b & 3);
          const float* img = dy + (b - (long long)(hh + top) * RO);      // this image's padded gradient
          acc = col_sum(row, w);
          if (mode_h == DLWP_PAD_WRAP) {
            if (hh >= H - top) acc += col_sum(img + (long long)(hh - (H - top)) * RO, w);
            if (hh < bottom) acc += col_sum(img + (long long)(top + H + hh) * RO, w);
          } else if (mode_h == DLWP_PAD_EDGE) {
            if (hh == 0)
              for (int rr = 0; rr < top; ++rr) acc += col_sum(img + (long long)rr * RO, w);
            if (hh == H - 1)
              for (int rr = top + H; rr < Ho; ++rr) acc += col_sum(img + (long long)rr * RO, w);
          } else if (mode_h >= DLWP_PAD_REFLECT) {
            for (int rr = 0; rr < top; ++rr)
              if (pad_map_col(rr - top, H, mode_h) == hh) acc += col_sum(img + (long long)rr * RO, w);
            for (int rr = top + H; rr < Ho; ++rr)
              if (pad_map_col(rr - top, H, mode_h) == hh) acc += col_sum(img + (long long)rr * RO, w);
          }
        }
        v[q] = acc;
      }
      float* dst = dx + r0 * RI + 4 * j;
      if (4 * j + 4 <= valid) {
        __builtin_nontemporal_store(v, (f32x4*)dst);
      } else {
        for (int q = 0; q < 4; ++q)
          if (4 * j + q < valid) dst[q] = v[q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// pad2d backward, gather form (rows too long for the LDS-staged kernel): every dx element sums the dy positions that
// were copies of it.
__global__ __launch_bounds__(256) void pad2d_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int outer,
                                                        int H, int W, int inner, int Ho, int Wo, int top, int bottom,
                                                        int left, int right, int mode_h, int mode_w) {
  const long long total = (long long)outer * H * W * inner;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % inner);
    long long q = i / inner;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H);
    const int o = (int)(q / H);
    // candidate padded rows / cols that map onto (h, w): the interior one, plus halo images
    int rows[2 + 1], nr = 0, r_lo = 0, r_hi = -1;  // explicit list for wrap, ranges for edge
    int cols[2 + 1], nc = 0, c_lo = 0, c_hi = -1;
    int r_lo2 = 0, r_hi2 = -1, c_lo2 = 0, c_hi2 = -1;
    rows[nr++] = h + top;
    if (mode_h == DLWP_PAD_WRAP) {
      if (h >= H - top) rows[nr++] = h - (H - top);
      if (h < bottom) rows[nr++] = top + H + h;
    } else if (mode_h == DLWP_PAD_EDGE) {
      if (h == 0) { r_lo = 0; r_hi = top - 1; }
      if (h == H - 1) { r_lo2 = top + H; r_hi2 = Ho - 1; }
    } else if (mode_h >= DLWP_PAD_REFLECT) {   // a row is the mirror image of at most one top and one bottom halo row
      const int a = mode_h == DLWP_PAD_REFLECT ? top - h : top - 1 - h;
      const int b = mode_h == DLWP_PAD_REFLECT ? top + 2 * H - 2 - h : top + 2 * H - 1 - h;
      if (a >= 0 && a < top) rows[nr++] = a;
      if (b >= top + H && b < Ho) rows[nr++] = b;
    }
    cols[nc++] = w + left;
    if (mode_w == DLWP_PAD_WRAP) {
      if (w >= W - left) cols[nc++] = w - (W - left);
      if (w < right) cols[nc++] = left + W + w;
    } else if (mode_w == DLWP_PAD_EDGE) {
      if (w == 0) { c_lo = 0; c_hi = left - 1; }
      if (w == W - 1) { c_lo2 = left + W; c_hi2 = Wo - 1; }
    } else if (mode_w >= DLWP_PAD_REFLECT) {
      const int a = mode_w == DLWP_PAD_REFLECT ? left - w : left - 1 - w;
      const int b = mode_w == DLWP_PAD_REFLECT ? left + 2 * W - 2 - w : left + 2 * W - 1 - w;
      if (a >= 0 && a < left) cols[nc++] = a;
      if (b >= left + W && b < Wo) cols[nc++] = b;
    }
    const float* base = dy + (long long)o * Ho * Wo * inner + ch;
    float acc = 0.f;
    auto col_sum = [&](int rr) {
      const float* rp = base + (long long)rr * Wo * inner;
      float s = 0.f;
      for (int k = 0; k < nc; ++k) s += rp[(long long)cols[k] * inner];
      for (int c = c_lo; c <= c_hi; ++c) s += rp[(long long)c * inner];
      for (int c = c_lo2; c <= c_hi2; ++c) s += rp[(long long)c * inner];
      return s;
    };
    for (int k = 0; k < nr; ++k) acc += col_sum(rows[k]);
    for (int r = r_lo; r <= r_hi; ++r) acc += col_sum(r);
    for (int r = r_lo2; r <= r_hi2; ++r) acc += col_sum(r);
    dx[i] = acc;
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// 2x2 max pooling / nearest up-sampling
// ------------------------------------------------------------------------------------------------------------------ //
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long long planes, int H, int W) {
  const int H2 = H / 2, W2 = W / 2;
  const long long total = planes * H2 * W2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % W2);
    const long long q = i / W2;
    const int r = (int)(q % H2);
    const long long p = q / H2;
    const float* s = x + (p * H + 2 * r) * W + 2 * j;
    y[i] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[W], s[W + 1]));
  }
}

// bfloat16 storage (config 4): the maximum of bf16 values is one of them, so the result is exact; a 2x2 window is two
// 32-bit loads when the row length is even
__global__ __launch_bounds__(256) void maxpool2_fwd_bf16_kernel(const unsigned short* __restrict__ x,
                                                                unsigned short* __restrict__ y, long long planes, int H,
                                                                int W) {
  const int H2 = H / 2, W2 = W / 2;
  const long long total = planes * H2 * W2;
  const bool even = (W & 1) == 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % W2);
    const long long q = i / W2;
    const int r = (int)(q % H2);
    const long long p = q / H2;
    const unsigned short* s = x + (p * H + 2 * r) * W + 2 * j;
    unsigned a, b;
    if (even) {
      a = *(const unsigned*)s;
      b = *(const unsigned*)(s + W);
    } else {
      a = (unsigned)s[0] | ((unsigned)s[1] << 16);
      b = (unsigned)s[W] | ((unsigned)s[W + 1] << 16);
    }
    const float m = fmaxf(fmaxf(__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xffff0000u)),
                          fmaxf(__builtin_bit_cast(float, b << 16), __builtin_bit_cast(float, b & 0xffff0000u)));
    y[i] = (unsigned short)(__builtin_bit_cast(unsigned, m) >> 16);
  }
}

// one thread per 2x2 window (incl. the partial windows of an odd edge, which receive zero gradient)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, long long planes, int H, int W) {
  const int H2 = H / 2, W2 = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
  const long long total = planes * Hc * Wc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % Wc);
    const long long q = i / Wc;
    const int r = (int)(q % Hc);
    const long long p = q / Hc;
    const long long o = (p * H + 2 * r) * W + 2 * j;
    if (r < H2 && j < W2) {
      const float a = x[o], b = x[o + 1], c = x[o + W], d = x[o + W + 1];
      const float g = dy[(p * H2 + r) * W2 + j];
      int arg = 0;  // first maximum in row-major window order
      float m = a;
      if (b > m) { m = b; arg = 1; }
      if (c > m) { m = c; arg = 2; }
      if (d > m) { m = d; arg = 3; }
      dx[o] = arg == 0 ? g : 0.f;
      dx[o + 1] = arg == 1 ? g : 0.f;
      dx[o + W] = arg == 2 ? g : 0.f;
      dx[o + W + 1] = arg == 3 ? g : 0.f;
    } else {
      const bool has_c = 2 * j + 1 < W, has_r = 2 * r + 1 < H;
      dx[o] = 0.f;
      if (has_c) dx[o + 1] = 0.f;
      if (has_r) dx[o + W] = 0.f;
      if (has_r && has_c) dx[o + W + 1] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            long long planes, int H, int W) {
  const long long total = planes * H * W;
  const int W2 = 2 * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % W);
    const long long q = i / W;  // = p*H + r
    const float v = x[i];
    float* d = y + (2 * q) * W2 + 2 * j;  // 8-byte aligned: (2q*2W + 2j) is even
    const f32x2 vv = {v, v};
    *(f32x2*)d = vv;
    *(f32x2*)(d + W2) = vv;
  }
}

// even W: a thread takes a PAIR of source elements (one 8-byte load) and writes two 16-byte row segments {a, a, b, b}
__global__ __launch_bounds__(256) void upsample2_fwd_pair_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 long long planes, int H, int W) {
  const int Wh = W / 2, W2 = 2 * W;
  const long long total = planes * H * Wh;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % Wh);
    const long long q = i / Wh;  // = p*H + r
    const f32x2 v = *(const f32x2*)(x + q * W + 2 * j);
    float* d = y + (2 * q) * W2 + 4 * j;   // 16-byte aligned: 2q*2W and 4j are multiples of 4
    const f32x4 vv = {v[0], v[0], v[1], v[1]};
    *(f32x4*)d = vv;
    *(f32x4*)(d + W2) = vv;
  }
}

// even W: two outputs per thread from two 16-byte loads
__global__ __launch_bounds__(256) void upsample2_bwd_pair_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                 long long planes, int H, int W) {
  const int Wh = W / 2, W2 = 2 * W;
  const long long total = planes * H * Wh;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % Wh);
    const long long q = i / Wh;
    const float* s = dy + (2 * q) * W2 + 4 * j;
    const f32x4 a = *(const f32x4*)s, b = *(const f32x4*)(s + W2);
    const f32x2 o = {(a[0] + a[1]) + (b[0] + b[1]), (a[2] + a[3]) + (b[2] + b[3])};   // same association as the scalar form
    *(f32x2*)(dx + q * W + 2 * j) = o;
  }
}

__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                            long long planes, int H, int W) {
  const long long total = planes * H * W;
  const int W2 = 2 * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % W);
    const long long q = i / W;
    const float* s = dy + (2 * q) * W2 + 2 * j;
    const f32x2 a = *(const f32x2*)s, b = *(const f32x2*)(s + W2);
    dx[i] = (a[0] + a[1]) + (b[0] + b[1]);
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// strided block copies
// ------------------------------------------------------------------------------------------------------------------ //
// copy runs of `run` contiguous floats; run (a, b, c) starts at src + a*s_sa + b*s_sb + c*s_sc and lands at
// dst + a*d_sa + b*d_sb + c*d_sc
template <int VEC>
__global__ __launch_bounds__(256) void copy_runs_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        long long na, long long nb, long long nc, long long run,
                                                        long long s_sa, long long s_sb, long long s_sc, long long d_sa,
                                                        long long d_sb, long long d_sc) {
  const long long per = run / VEC;
  const long long total = na * nb * nc * per;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long e = (i % per) * VEC;
    long long k = i / per;
    const long long c = k % nc;
    k /= nc;
    const long long b = k % nb, a = k / nb;
    const float* s = src + a * s_sa + b * s_sb + c * s_sc + e;
    float* d = dst + a * d_sa + b * d_sb + c * d_sc + e;
    if (VEC == 4) *(f32x4*)d = *(const f32x4*)s;
    else *d = *s;
  }
}


extern "C" {

// the flat row-group kernels: rows per wave so that a group is >= ~2.5 KB (loads in flight per wave), and the multiply-high
// constant of a division by d (exact for n * d < 2^32: n < 16 rows x 4096 floats)
static inline int pad_flat_rows(int row_floats) { return row_floats <= 80 ? 16 : (row_floats <= 160 ? 8 : 4); }
static inline unsigned pad_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

int dlwp_pad2d_fwd(dlwp_handle_t h, const void* x, void* y, int outer, int H, int W, int inner, dlwp_pad2d p, int dtype,
                   void* stream) {
  DLWP_TAPE(h, stream, dlwp_pad2d_fwd, h, x, y, outer, H, W, inner, p, dtype);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_pad2d_fwd: null handle");
  if (outer == 0) return DLWP_OK;  // empty batch: torch hands out null data pointers for empty tensors
  DLWP_CHECK_ARG(x && y, "dlwp_pad2d_fwd: null pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_pad2d_fwd: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(outer >= 0 && H > 0 && W > 0 && inner > 0, "dlwp_pad2d_fwd: bad shape");
  DLWP_CHECK_ARG(p.top >= 0 && p.bottom >= 0 && p.left >= 0 && p.right >= 0, "dlwp_pad2d_fwd: negative padding");
  DLWP_CHECK_ARG((unsigned)p.mode_h <= 4u && (unsigned)p.mode_w <= 4u, "dlwp_pad2d_fwd: unknown pad mode");
  // the reference's slices do not tile (custom.py:197-200): wrap padding larger than the axis is an error there too;
  // tf.pad rejects REFLECT amounts >= the axis and SYMMETRIC amounts > the axis
  DLWP_CHECK_ARG(p.mode_h != DLWP_PAD_WRAP || (p.top <= H && p.bottom <= H),
                 "dlwp_pad2d_fwd: periodic row padding (%d,%d) exceeds H=%d", p.top, p.bottom, H);
  DLWP_CHECK_ARG(p.mode_w != DLWP_PAD_WRAP || (p.left <= W && p.right <= W),
                 "dlwp_pad2d_fwd: periodic column padding (%d,%d) exceeds W=%d", p.left, p.right, W);
  DLWP_CHECK_ARG(dlwp_pad_fits(p.top, p.bottom, H, p.mode_h) && dlwp_pad_fits(p.left, p.right, W, p.mode_w),
                 "dlwp_pad2d_fwd: mirror padding exceeds the axis");
  if (outer == 0) return DLWP_OK;
  const int Ho = H + p.top + p.bottom, Wo = W + p.left + p.right;
  const long long RI = (long long)W * inner, RO = (long long)Wo * inner;
  if (inner == 1 && aligned16(x) && aligned16(y) && Wo <= 4096) {      // r5: the flat row-group kernel (any row length)
    const int rows = pad_flat_rows(W < Wo ? W : Wo);
    const int flat_lds = (int)((RI + 6) / 4 * 4 + 4);
    const size_t bytes = (size_t)flat_lds * 4 * rows * sizeof(float) + (size_t)4 * rows * sizeof(long long);
    if (bytes <= 64 * 1024 && bytes <= (size_t)h->lds_bytes) {
      const long long groups = ((long long)outer * Ho + rows - 1) / rows;
      int grid = (int)((groups + 3) / 4);
      const int cap = h->cu_count * 8;
      if (grid > cap) grid = cap;
      const unsigned m_nv = pad_magic((unsigned)((RI + 6) >> 2)), m_ro = pad_magic((unsigned)RO);
#define PADF_LAUNCH(R, MH, MW)                                                                                              \
  pad2d_fwd_flat_kernel<R, MH, MW><<<grid, 256, bytes, (hipStream_t)stream>>>((const float*)x, (float*)y, outer, H, W, Ho, Wo,   \
                                                                              p.top, p.left, p.mode_h, p.mode_w, flat_lds, m_nv, m_ro)
      if (p.mode_h == DLWP_PAD_ZERO && p.mode_w == DLWP_PAD_WRAP) {        // PeriodicPadding2D((0, k)) + ZeroPadding2D((k, 0))
        if (rows == 16) PADF_LAUNCH(16, DLWP_PAD_ZERO, DLWP_PAD_WRAP);
        else if (rows == 8) PADF_LAUNCH(8, DLWP_PAD_ZERO, DLWP_PAD_WRAP);
        else PADF_LAUNCH(4, DLWP_PAD_ZERO, DLWP_PAD_WRAP);
      } else {
        if (rows == 16) PADF_LAUNCH(16, -1, -1);
        else if (rows == 8) PADF_LAUNCH(8, -1, -1);
        else PADF_LAUNCH(4, -1, -1);
      }
#undef PADF_LAUNCH
      DLWP_LAUNCH_CHECK("pad2d_fwd_flat_kernel");
      return DLWP_OK;
    }
  }
  const int row_lds = (int)((RI + 3) / 4 * 4);
  const size_t lds_bytes = (size_t)row_lds * 4 * 4 * sizeof(float);  // 4 waves x ROWS(4) row slots
  DLWP_CHECK_ARG(lds_bytes <= (size_t)h->lds_bytes, "dlwp_pad2d_fwd: row of %lld floats does not fit in LDS", RI);
  const long long n_rows = (long long)outer * Ho;
  int grid = (int)((n_rows + 15) / 16);
  const int cap = h->cu_count * 8;
  if (grid > cap) grid = cap;
  const bool vec = (RI % 4 == 0) && (RO % 4 == 0) && aligned16(x) && aligned16(y);
  const bool vec2 = (RI % 2 == 0) && (RO % 2 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 7) == 0;
  hipStream_t s = (hipStream_t)stream;
#define PAD_LAUNCH(V, I1)                                                                                            \
  pad2d_fwd_kernel<V, I1><<<grid, 256, lds_bytes, s>>>((const float*)x, (float*)y, outer, H, W, inner, Ho, Wo, p.top, \
                                                       p.left, p.mode_h, p.mode_w, row_lds)
  if (inner == 1) {
    if (vec) PAD_LAUNCH(4, true);
    else if (vec2) PAD_LAUNCH(2, true);
    else PAD_LAUNCH(1, true);
  } else {
    if (vec) PAD_LAUNCH(4, false);
    else if (vec2) PAD_LAUNCH(2, false);
    else PAD_LAUNCH(1, false);
  }
#undef PAD_LAUNCH
  DLWP_LAUNCH_CHECK("pad2d_fwd_kernel");
  return DLWP_OK;
}

int dlwp_pad2d_bwd(dlwp_handle_t h, const void* dy, void* dx, int outer, int H, int W, int inner, dlwp_pad2d p,
                   int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_pad2d_bwd, h, dy, dx, outer, H, W, inner, p, dtype);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_pad2d_bwd: null handle");
  if (outer == 0) return DLWP_OK;
  DLWP_CHECK_ARG(dy && dx, "dlwp_pad2d_bwd: null pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_pad2d_bwd: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(outer >= 0 && H > 0 && W > 0 && inner > 0, "dlwp_pad2d_bwd: bad shape");
  DLWP_CHECK_ARG(p.top >= 0 && p.bottom >= 0 && p.left >= 0 && p.right >= 0, "dlwp_pad2d_bwd: negative padding");
  DLWP_CHECK_ARG((unsigned)p.mode_h <= 4u && (unsigned)p.mode_w <= 4u, "dlwp_pad2d_bwd: unknown pad mode");
  DLWP_CHECK_ARG(p.mode_h != DLWP_PAD_WRAP || (p.top <= H && p.bottom <= H), "dlwp_pad2d_bwd: periodic rows exceed H");
  DLWP_CHECK_ARG(p.mode_w != DLWP_PAD_WRAP || (p.left <= W && p.right <= W), "dlwp_pad2d_bwd: periodic cols exceed W");
  DLWP_CHECK_ARG(dlwp_pad_fits(p.top, p.bottom, H, p.mode_h) && dlwp_pad_fits(p.left, p.right, W, p.mode_w),
                 "dlwp_pad2d_bwd: mirror padding exceeds the axis");
  if (outer == 0) return DLWP_OK;
  if (inner == 1 && aligned16(dy) && aligned16(dx) && W + p.left + p.right <= 4096) {      // r5: the flat row-group kernel
    const int Ho = H + p.top + p.bottom, Wo = W + p.left + p.right;
    const int rows = pad_flat_rows(W);
    const int flat_lds = (int)((Wo + 6) / 4 * 4 + 4);
    const size_t bytes = (size_t)flat_lds * 4 * rows * sizeof(float) + (size_t)4 * rows * (sizeof(long long) + sizeof(int));
    if (bytes <= 64 * 1024 && bytes <= (size_t)h->lds_bytes) {
      const long long groups = ((long long)outer * H + rows - 1) / rows;
      int grid = (int)((groups + 3) / 4);
      const int cap = h->cu_count * 8;
      if (grid > cap) grid = cap;
      const unsigned m_nv = pad_magic((unsigned)((Wo + 6) >> 2)), m_ri = pad_magic((unsigned)W);
#define PADBF_LAUNCH(R, MH, MW)                                                                                              \
  pad2d_bwd_flat_kernel<R, MH, MW><<<grid, 256, bytes, (hipStream_t)stream>>>((const float*)dy, (float*)dx, outer, H, W, Ho, Wo,  \
                                                                              p.top, p.bottom, p.left, p.right, p.mode_h, p.mode_w, \
                                                                              flat_lds, m_nv, m_ri)
      if (p.mode_h == DLWP_PAD_ZERO && p.mode_w == DLWP_PAD_WRAP) {        // PeriodicPadding2D((0, k)) + ZeroPadding2D((k, 0))
        if (rows == 16) PADBF_LAUNCH(16, DLWP_PAD_ZERO, DLWP_PAD_WRAP);
        else if (rows == 8) PADBF_LAUNCH(8, DLWP_PAD_ZERO, DLWP_PAD_WRAP);
        else PADBF_LAUNCH(4, DLWP_PAD_ZERO, DLWP_PAD_WRAP);
      } else {
        if (rows == 16) PADBF_LAUNCH(16, -1, -1);
        else if (rows == 8) PADBF_LAUNCH(8, -1, -1);
        else PADBF_LAUNCH(4, -1, -1);
      }
#undef PADBF_LAUNCH
      DLWP_LAUNCH_CHECK("pad2d_bwd_flat_kernel");
      return DLWP_OK;
    }
  }
  {
    // row-staged kernel whenever 4 waves x 4 padded rows fit in LDS (every shape of the reference's networks does)
    const int Ho = H + p.top + p.bottom, Wo = W + p.left + p.right;
    const long long RI = (long long)W * inner, RO = (long long)Wo * inner;
    const int row_lds = (int)((RO + 3) / 4 * 4);
    const size_t lds_bytes = (size_t)row_lds * 4 * 4 * sizeof(float);
    if (lds_bytes <= (size_t)h->lds_bytes && lds_bytes <= 64 * 1024) {
      const long long n_rows = (long long)outer * H;
      int grid = (int)((n_rows + 15) / 16);
      const int cap = h->cu_count * 8;
      if (grid > cap) grid = cap;
      const bool vec = (RI % 4 == 0) && (RO % 4 == 0) && aligned16(dy) && aligned16(dx);
      const bool vec2 = (RI % 2 == 0) && (RO % 2 == 0) && ((((uintptr_t)dy) | ((uintptr_t)dx)) & 7) == 0;
      hipStream_t s = (hipStream_t)stream;
#define PADB_LAUNCH(V, I1)                                                                                              \
  pad2d_bwd_rows_kernel<V, I1><<<grid, 256, lds_bytes, s>>>((const float*)dy, (float*)dx, outer, H, W, inner, Ho, Wo,   \
                                                            p.top, p.bottom, p.left, p.right, p.mode_h, p.mode_w, row_lds)
      if (inner == 1) {
        if (vec) PADB_LAUNCH(4, true);
        else if (vec2) PADB_LAUNCH(2, true);
        else PADB_LAUNCH(1, true);
      } else {
        if (vec) PADB_LAUNCH(4, false);
        else if (vec2) PADB_LAUNCH(2, false);
        else PADB_LAUNCH(1, false);
      }
#undef PADB_LAUNCH
      DLWP_LAUNCH_CHECK("pad2d_bwd_rows_kernel");
      return DLWP_OK;
    }
  }
  const long long total = (long long)outer * H * W * inner;
  pad2d_bwd_kernel<<<grid_for(total, 256, h->cu_count), 256, 0, (hipStream_t)stream>>>(
      (const float*)dy, (float*)dx, outer, H, W, inner, H + p.top + p.bottom, W + p.left + p.right, p.top, p.bottom,
      p.left, p.right, p.mode_h, p.mode_w);
  DLWP_LAUNCH_CHECK("pad2d_bwd_kernel");
  return DLWP_OK;
}

#define POOL_ARGS_OK(name)                                                                  \
  DLWP_CHECK_ARG(h != nullptr, name ": null handle");                                       \
  if (xs.n == 0) return DLWP_OK;                                                            \
  DLWP_CHECK_ARG(x_ok, name ": null pointer");                                              \
  DLWP_CHECK_ARG(dtype == DLWP_F32, name ": dtype %d not supported", dtype);                \
  DLWP_CHECK_ARG(xs.n >= 0 && xs.c > 0 && xs.h > 0 && xs.w > 0, name ": bad shape")

int dlwp_maxpool2_fwd(dlwp_handle_t h, const void* x, void* y, dlwp_shape4 xs, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_maxpool2_fwd, h, x, y, xs, dtype);
  const bool x_ok = x && y;
  const bool bf16 = dtype == DLWP_BF16;
  if (bf16) dtype = DLWP_F32;  // the shared argument check knows fp32 only; this entry point also stores bf16
  POOL_ARGS_OK("dlwp_maxpool2_fwd");
  const long long planes = (long long)xs.n * xs.c;
  const long long total = planes * (xs.h / 2) * (xs.w / 2);
  if (total == 0) return DLWP_OK;
  if (bf16) {
    maxpool2_fwd_bf16_kernel<<<grid_for(total, 256, h->cu_count), 256, 0, (hipStream_t)stream>>>(
        (const unsigned short*)x, (unsigned short*)y, planes, xs.h, xs.w);
    DLWP_LAUNCH_CHECK("maxpool2_fwd_bf16_kernel");
    return DLWP_OK;
  }
  maxpool2_fwd_kernel<<<grid_for(total, 256, h->cu_count), 256, 0, (hipStream_t)stream>>>((const float*)x, (float*)y,
                                                                                          planes, xs.h, xs.w);
  DLWP_LAUNCH_CHECK("maxpool2_fwd_kernel");
  return DLWP_OK;
}

int dlwp_maxpool2_bwd(dlwp_handle_t h, const void* x, const void* dy, void* dx, dlwp_shape4 xs, int dtype,
                      void* stream) {
  DLWP_TAPE(h, stream, dlwp_maxpool2_bwd, h, x, dy, dx, xs, dtype);
  const bool x_ok = x && dy && dx;
  POOL_ARGS_OK("dlwp_maxpool2_bwd");
  const long long planes = (long long)xs.n * xs.c;
  const long long total = planes * ((xs.h + 1) / 2) * ((xs.w + 1) / 2);
  if (total == 0) return DLWP_OK;
  maxpool2_bwd_kernel<<<grid_for(total, 256, h->cu_count), 256, 0, (hipStream_t)stream>>>(
      (const float*)x, (const float*)dy, (float*)dx, planes, xs.h, xs.w);
  DLWP_LAUNCH_CHECK("maxpool2_bwd_kernel");
  return DLWP_OK;
}

int dlwp_upsample2_fwd(dlwp_handle_t h, const void* x, void* y, dlwp_shape4 xs, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_upsample2_fwd, h, x, y, xs, dtype);
  const bool x_ok = x && y;
  POOL_ARGS_OK("dlwp_upsample2_fwd");
  const long long planes = (long long)xs.n * xs.c;
  const long long total = planes * xs.h * xs.w;
  if (total == 0) return DLWP_OK;
  DLWP_CHECK_ARG((((uintptr_t)y) & 7) == 0, "dlwp_upsample2_fwd: output must be 8-byte aligned");
  if ((xs.w & 1) == 0 && aligned16(y) && (((uintptr_t)x) & 7) == 0)
    upsample2_fwd_pair_kernel<<<grid_for(total / 2, 256, h->cu_count), 256, 0, (hipStream_t)stream>>>(
        (const float*)x, (float*)y, planes, xs.h, xs.w);
  else
    upsample2_fwd_kernel<<<grid_for(total, 256, h->cu_count), 256, 0, (hipStream_t)stream>>>((const float*)x, (float*)y,
                                                                                             planes, xs.h, xs.w);
  DLWP_LAUNCH_CHECK("upsample2_fwd_kernel");
  return DLWP_OK;
}

// xs = shape of dx (the low-resolution tensor); dy is (n, c, 2h, 2w)
int dlwp_upsample2_bwd(dlwp_handle_t h, const void* dy, void* dx, dlwp_shape4 xs, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_upsample2_bwd, h, dy, dx, xs, dtype);
  const bool x_ok = dy && dx;
  POOL_ARGS_OK("dlwp_upsample2_bwd");
  const long long planes = (long long)xs.n * xs.c;
  const long long total = planes * xs.h * xs.w;
  if (total == 0) return DLWP_OK;
  DLWP_CHECK_ARG((((uintptr_t)dy) & 7) == 0, "dlwp_upsample2_bwd: dy must be 8-byte aligned");
  if ((xs.w & 1) == 0 && aligned16(dy) && (((uintptr_t)dx) & 7) == 0)
    upsample2_bwd_pair_kernel<<<grid_for(total / 2, 256, h->cu_count), 256, 0, (hipStream_t)stream>>>(
        (const float*)dy, (float*)dx, planes, xs.h, xs.w);
  else
    upsample2_bwd_kernel<<<grid_for(total, 256, h->cu_count), 256, 0, (hipStream_t)stream>>>((const float*)dy, (float*)dx,
                                                                                             planes, xs.h, xs.w);
  DLWP_LAUNCH_CHECK("upsample2_bwd_kernel");
  return DLWP_OK;
}

int dlwp_copy_channels(dlwp_handle_t h, const void* src, void* dst, int n, int c, int hw, int src_c_off, int src_c_total,
                       int dst_c_off, int dst_c_total, int dtype, void* stream) {
  DLWP_TAPE(h, stream, dlwp_copy_channels, h, src, dst, n, c, hw, src_c_off, src_c_total, dst_c_off, dst_c_total, dtype);
  DLWP_CHECK_ARG(h != nullptr, "dlwp_copy_channels: null handle");
  if (n == 0) return DLWP_OK;
  DLWP_CHECK_ARG(src && dst, "dlwp_copy_channels: null pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_copy_channels: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(n >= 0 && c > 0 && hw > 0, "dlwp_copy_channels: bad shape");
  DLWP_CHECK_ARG(src_c_off >= 0 && src_c_off + c <= src_c_total, "dlwp_copy_channels: source channel range");
  DLWP_CHECK_ARG(dst_c_off >= 0 && dst_c_off + c <= dst_c_total, "dlwp_copy_channels: destination channel range");
  if (n == 0) return DLWP_OK;
  const long long run = (long long)c * hw;  // channels are adjacent: one contiguous run per sample
  const float* s = (const float*)src + (long long)src_c_off * hw;
  float* d = (float*)dst + (long long)dst_c_off * hw;
  const long long s_sa = (long long)src_c_total * hw, d_sa = (long long)dst_c_total * hw;
  const bool vec = run % 4 == 0 && s_sa % 4 == 0 && d_sa % 4 == 0 && aligned16(s) && aligned16(d);
  const long long items = (long long)n * (vec ? run / 4 : run);
  const int grid = grid_for(items, 256, h->cu_count);
  if (vec)
    copy_runs_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>(s, d, n, 1, 1, run, s_sa, 0, 0, d_sa, 0, 0);
  else
    copy_runs_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(s, d, n, 1, 1, run, s_sa, 0, 0, d_sa, 0, 0);
  DLWP_LAUNCH_CHECK("copy_runs_kernel");
  return DLWP_OK;
}

int dlwp_series_merge_time(dlwp_handle_t h, const void* series, void* out, int t, int n, int time_dim, int v, int hw,
                           int dtype, void* stream) {
  DLWP_UNTAPED(dlwp_series_merge_time);
  DLWP_CHECK_ARG(h && series && out, "dlwp_series_merge_time: null handle or pointer");
  DLWP_CHECK_ARG(dtype == DLWP_F32, "dlwp_series_merge_time: dtype %d not supported", dtype);
  DLWP_CHECK_ARG(t >= 0 && n >= 0 && time_dim > 0 && v > 0 && hw > 0, "dlwp_series_merge_time: bad shape");
  if (t == 0 || n == 0) return DLWP_OK;
  // in [t, n, td, run] with run = v*hw  ->  out [t, td, n, run]
  const long long run = (long long)v * hw;
  const bool vec = run % 4 == 0 && aligned16(series) && aligned16(out);
  const long long items = (long long)t * n * time_dim * (vec ? run / 4 : run);
  const int grid = grid_for(items, 256, h->cu_count);
  const long long slot = (long long)n * time_dim * run;
  if (vec)
    copy_runs_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)series, (float*)out, t, n, time_dim, run,
                                                               slot, time_dim * run, run, slot, run, (long long)n * run);
  else
    copy_runs_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)series, (float*)out, t, n, time_dim, run,
                                                               slot, time_dim * run, run, slot, run, (long long)n * run);
  DLWP_LAUNCH_CHECK("copy_runs_kernel");
  return DLWP_OK;
}

}  // extern "C"
