// Device-side building blocks shared by the axvs kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace axvs {

typedef unsigned short u16;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;     // CDNA wavefront
constexpr int kBlk = 32;      // K-block width of the "blocked" 16-bit activation/weight layout

// 16-bit MFMA operand type: fp16 (BF=false) or bf16 (BF=true).  Accumulation is always fp32.
template <bool BF>
struct H16;

template <>
struct H16<false> {
  static __device__ __forceinline__ u16 from_f32(float x) { return __builtin_bit_cast(u16, (_Float16)x); }
  static __device__ __forceinline__ float to_f32(u16 x) { return (float)__builtin_bit_cast(_Float16, x); }
  // D[i][j] += sum_k A[i][k] * B[k][j];  lane l holds A[l&15][8(l>>4)+e], B[8(l>>4)+e][l&15], D[4(l>>4)+r][l&15]
  static __device__ __forceinline__ f32x4 mfma(u16x8 a, u16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

template <>
struct H16<true> {
  static __device__ __forceinline__ u16 from_f32(float x) { return __builtin_bit_cast(u16, (__bf16)x); }
  static __device__ __forceinline__ float to_f32(u16 x) { return __uint_as_float(((unsigned)x) << 16); }
  static __device__ __forceinline__ f32x4 mfma(u16x8 a, u16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

// ---- 16-bit VALU helper for the reassociated temporal half (axvs_fused.h): acc += a * x on 8 packed elements.
//      fp16: v_pk_fma_f16; bf16 (non-default operand type): fp32 math.  (The per-row dot products of that half run on the
//      matrix pipe since round 4: diagonal of a 16 x 16 x 32 MFMA.)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
struct H2x4 { h16x2 v[4]; };

// fp32-accumulated dot product of 8 packed 16-bit pairs: four v_dot2_f32_f16 (v_dot2_f32_bf16).  Used by the hybrid form of the temporal
// logits (axvs_fused.h, AXVS_LOGITS_VALU_KB): some channel blocks on the VALU beside the MFMA diagonals of the others.
template <bool BF>
__device__ __forceinline__ float dot8_acc(u16x8 a, u16x8 x, float acc) {
  if constexpr (!BF) {
    const H2x4 as = __builtin_bit_cast(H2x4, a), xs = __builtin_bit_cast(H2x4, x);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2(as.v[i], xs.v[i], acc, false);
    return acc;
  } else {
    typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
    struct B2x4 { b16x2 v[4]; };
    const B2x4 as = __builtin_bit_cast(B2x4, a), xs = __builtin_bit_cast(B2x4, x);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(as.v[i], xs.v[i], acc, false);
    return acc;
  }
}

// acc (8 packed 16-bit values) += a * x.  fp16: four v_pk_fma_f16 (running sum kept in fp16: the result is an MFMA operand and
// would be rounded to 16 bits anyway; T <= 5 terms).  bf16: fp32 math, rounded per call.
template <bool BF>
__device__ __forceinline__ u16x8 axpy8(float a, u16x8 x, u16x8 acc) {
  if constexpr (!BF) {
    const H2x4 xs = __builtin_bit_cast(H2x4, x);
    H2x4 as = __builtin_bit_cast(H2x4, acc);
    const h16x2 a2 = {(_Float16)a, (_Float16)a};
#pragma unroll
    for (int i = 0; i < 4; ++i) as.v[i] = xs.v[i] * a2 + as.v[i];
    return __builtin_bit_cast(u16x8, as);
  } else {
    u16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = H16<BF>::from_f32(H16<BF>::to_f32(acc[i]) + a * H16<BF>::to_f32(x[i]));
    return r;
  }
}

template <bool BF>
__device__ __forceinline__ u16x8 cvt8(const float (&v)[8]) {
  u16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = H16<BF>::from_f32(v[i]);
  return r;
}

template <bool BF>
__device__ __forceinline__ u16x4 cvt4(f32x4 v) {
  u16x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = H16<BF>::from_f32(v[i]);
  return r;
}

// Sequence-order row m' = s*N + n  ->  row of the natural [B,T,H,W] token grid.
//   s = b*Loff + o (o = off-axis coordinate), n = t*L + l (l = on-axis coordinate)
//   nat = b*sB + t*sT + l*sL + o*sO           (rows; multiply by C for elements)
// height pass: L=H, Loff=W, sL=W, sO=1.  width pass: L=W, Loff=H, sL=1, sO=W.
// identity ([S,N,C] input): Loff=1, sB=N, sT=L, sL=1, sO=0.
// PADDED FRAMES (round 5; the fused trajectory tier only): a frame whose length is not a multiple of 16 (the shipped VIPSeg maps:
// 49 x 85, 25 x 43) occupies L = roundup16(Lv) rows of the sequence-order row space, of which the first Lv exist; rows l >= Lv are
// clamped copies of row Lv - 1 that are computed and never stored (like the rows past a sequence's end).  Every frame then starts at
// a multiple of 16 rows, so a 16-row MFMA tile never straddles two frames: K rows and V^T fragments are stored 16 / 8 bytes per lane
// for ANY frame length, and the row tiles of a pass can hand K / V^T over inside one launch.  Lv == 0: every row exists (L = Lv).
struct RowMap {
  int N, L, Loff;
  long long sB, sT, sL, sO;
  int Lv;
};

__device__ __forceinline__ long long nat_row(const RowMap& rm, int mp) {
  int s = mp / rm.N, n = mp - s * rm.N;
  int b = s / rm.Loff, o = s - b * rm.Loff;
  int t = n / rm.L, l = n - t * rm.L;
  if (rm.Lv) l = min(l, rm.Lv - 1);
  return b * rm.sB + t * rm.sT + l * rm.sL + o * rm.sO;
}
// does sequence-order row mp exist (padded frames: rows l >= Lv of a frame do not)?
__device__ __forceinline__ bool row_exists(const RowMap& rm, int mp) {
  if (!rm.Lv) return true;
  const int n = mp % rm.N;
  return n % rm.L < rm.Lv;
}

// Rows of a [frames, hw, C] tensor whose frames are `hw + extra` rows apart (a level of the pixel decoder's concatenated token
// buffer, used in place): natural row m -> row index in the buffer.  hw = 0: contiguous rows.
struct RowStride {
  int hw;
  long long extra;
  __device__ __forceinline__ long long row(long long m) const { return hw ? m + (m / hw) * extra : m; }
};

// exact (erf) GELU: F.gelu / nn.GELU() defaults
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// ---- rows of pixels: their count is whatever the image size gives (193 x 337 maps), so a row may start at any 4-byte boundary
//      and end inside a group of four.  Alignment-aware 4-float accesses for the kernels that walk such rows.
// row alignment (floats) of an operand whose rows start at base + i * ld + j * step for integers i, j
inline int row_align(const void* base, long long ld, long long step) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  if (a % 16 == 0 && ld % 4 == 0 && step % 4 == 0) return 4;
  if (a % 8 == 0 && ld % 2 == 0 && step % 2 == 0) return 2;
  return 1;
}

// four consecutive floats at p of which the first `valid` exist (<= 0: none); al: alignment of p in floats.
// GLOBAL memory only: gfx950 code objects run with unaligned access enabled, so four floats at a 4-byte boundary are ONE
// global_load_dwordx4 / global_store_dwordx4 (the type below tells the compiler the alignment; it emits the wide instruction) --
// rows of 130082 pixels (the shipped VIPSeg training shape: 8-byte aligned) move 16 bytes per lane like aligned ones.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 ldg4(const float* p, int valid, int /*al*/) {
  if (valid >= 4) {
    const f32x4_u v = *reinterpret_cast<const f32x4_u*>(p);
    return float4{v.x, v.y, v.z, v.w};
  }
  float4 r = {0.f, 0.f, 0.f, 0.f};
  if (valid > 0) r.x = p[0];
  if (valid > 1) r.y = p[1];
  if (valid > 2) r.z = p[2];
  return r;
}
__device__ __forceinline__ void stg4(float* p, float4 v, int valid, int /*al*/) {
  if (valid >= 4) {
    *reinterpret_cast<f32x4_u*>(p) = f32x4_u{v.x, v.y, v.z, v.w};
    return;
  }
  if (valid > 0) p[0] = v.x;
  if (valid > 1) p[1] = v.y;
  if (valid > 2) p[2] = v.z;
}

// Same decomposition, also returning the (t, h, w) grid coordinates packed as t | h << 8 | w << 20 (T <= 255, H, W <= 4095):
// what an in-kernel positional embedding needs.  l_is_h: the on-axis coordinate l is the image row (height pass).
__device__ __forceinline__ long long nat_row_coords(const RowMap& rm, int mp, int l_is_h, int* packed) {
  int s = mp / rm.N, n = mp - s * rm.N;
  int b = s / rm.Loff, o = s - b * rm.Loff;
  int t = n / rm.L, l = n - t * rm.L;
  if (rm.Lv) l = min(l, rm.Lv - 1);
  *packed = t | ((l_is_h ? l : o) << 8) | ((l_is_h ? o : l) << 20);
  return b * rm.sB + t * rm.sT + l * rm.sL + o * rm.sO;
}

// PositionEmbeddingSine3D (WC/pos_embeddings.py:86-130, mask = None) evaluated inside a loader instead of read from HBM:
//   pos[t,h,w,c] = (c < n ? sincos(y_h / dim_t(c)) : sincos(x_w / dim_t(c - n))) + sincos(z_t / dim_tz(c)) [+ level[c]],  n = C/2,
//   dim_t(i) = temperature^(2 floor(i/2) / n), dim_tz(i) = temperature^(2 floor(i/2) / C), sin for even channels, cos for odd ones,
//   coordinates 1..L, scaled by scale / (L + 1e-6) when normalised.
struct PosGen {
  int mode;               // 0: read the pos tensor, 1: generate
  int l_is_h;             // RowMap's on-axis coordinate is the image row
  float zs, ys, xs;       // coordinate scale factors (1 when not normalised)
  float ke_yx, ke_z;      // -log2(temperature) * 2 / n  and  -log2(temperature) * 2 / C
  int n;                  // C / 2
  const float* level;     // nullable fp32 [C]
};

// the 4 consecutive channels c0..c0+3 (c0 % 4 == 0) of one token
struct PosGenLane {
  float f0, f1, g0, g1;   // yx / z frequencies of the two channel pairs, in revolutions per coordinate unit
  float lv[4];
  int xside;
  __device__ __forceinline__ void init(const PosGen& pg, int c0) {
    const float inv2pi = 0.15915494309189535f;
    xside = c0 >= pg.n;
    const int cc = c0 - (xside ? pg.n : 0);
    f0 = __builtin_amdgcn_exp2f(pg.ke_yx * (float)(cc >> 1)) * inv2pi;
    f1 = __builtin_amdgcn_exp2f(pg.ke_yx * (float)((cc >> 1) + 1)) * inv2pi;
    g0 = __builtin_amdgcn_exp2f(pg.ke_z * (float)(c0 >> 1)) * inv2pi;
    g1 = __builtin_amdgcn_exp2f(pg.ke_z * (float)((c0 >> 1) + 1)) * inv2pi;
    if (pg.level) {
      const float4 l = *reinterpret_cast<const float4*>(pg.level + c0);
      lv[0] = l.x; lv[1] = l.y; lv[2] = l.z; lv[3] = l.w;
    } else {
      lv[0] = lv[1] = lv[2] = lv[3] = 0.f;
    }
  }
  // v_sin_f32 / v_cos_f32 take revolutions and need a bounded argument: reduce with v_fract_f32 first
  __device__ __forceinline__ float4 eval(const PosGen& pg, int packed) const {
    const float z = (float)((packed & 255) + 1) * pg.zs;
    const float u = xside ? (float)(((packed >> 20) & 4095) + 1) * pg.xs : (float)(((packed >> 8) & 4095) + 1) * pg.ys;
    const float a0 = __builtin_amdgcn_fractf(u * f0), a1 = __builtin_amdgcn_fractf(u * f1);
    const float b0 = __builtin_amdgcn_fractf(z * g0), b1 = __builtin_amdgcn_fractf(z * g1);
    return float4{__builtin_amdgcn_sinf(a0) + __builtin_amdgcn_sinf(b0) + lv[0], __builtin_amdgcn_cosf(a0) + __builtin_amdgcn_cosf(b0) + lv[1],
                  __builtin_amdgcn_sinf(a1) + __builtin_amdgcn_sinf(b1) + lv[2], __builtin_amdgcn_cosf(a1) + __builtin_amdgcn_cosf(b1) + lv[3]};
  }
};

// Blocked 16-bit matrix [K/32][R][32]: element (row r, col k).
__device__ __forceinline__ long long blk_off(long long R, long long r, int k) {
  return ((long long)(k >> 5) * R + r) * kBlk + (k & 31);
}

// Blocked 16-bit WEIGHT matrix [K/32][Rp/16][4 chunks][16 rows][8] (Rp = R rounded up to 16): the 1 KiB of a (k-block, 16 rows)
// block is stored in MFMA-FRAGMENT order, so that lane (i = lane & 15, g = lane >> 4) of a fragment load -- row i, 8 k-values of
// chunk g -- reads bytes [16 lane, 16 lane + 16) of the block: consecutive lanes, consecutive addresses.  With plain 64-byte rows
// ([K/32][R][32], the activation layout above) the same load makes 4 consecutive lanes touch 4 different 64-byte segments, and the
// L2 -> CU weight stream of the N-split kernels ran ~ 1.5x slower for it (tools/hw/ffn_overlap.hip: 3.9 k vs 2.6 k cycles per FFN
// phase).  Every packer (axvs_misc.h) writes and every weight reader (w_frag, the GEMM kernels) addresses through this function.
__device__ __forceinline__ long long wblk_off(long long R, long long r, int k) {
  const long long Rp = (R + 15) & ~15ll;
#ifdef AXVS_WEIGHT_ROWS   // diagnostic (tools/ab_variants.py): rounds 1-3's plain 64-byte rows, for same-box A/B runs of the access pattern
                          // (w_frag_u / ffn_wide_kernel assume the fragment order: run such a build with option ffn_wide = 2)
  return ((long long)(k >> 5) * Rp + r) * kBlk + (k & 31);
#endif
  return ((long long)(k >> 5) * Rp + (r & ~15ll)) * kBlk + ((k >> 3) & 3) * 128 + (r & 15) * 8 + (k & 7);
}

// LDS image of a [rows][32] 16-bit tile (64-byte rows) read as MFMA fragments with ds_read_b128:
// lane (i = lane&15, g = lane>>4) reads the 16-byte chunk g of row i.  XOR-ing the chunk index with
// kSwz[(row>>2)&3] makes the four 16-lane ds_read_b128 groups hit 16 distinct 16-byte slots.
__device__ __forceinline__ int swz_chunk(int row, int g) {
  // f = [0,3,2,1]  ==  (4 - h) & 3
  return g ^ ((4 - ((row >> 2) & 3)) & 3);
}

// ---- cross-lane exchange on the VALU (DPP / v_permlane*_swap): no trip through the LDS crossbar, no lgkmcnt ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
// value of lane ^ m for m in {1, 2, 4*, 8*, 16, 32}.  (* within the butterfly order 1,2,4,8: after the xor-1 and
// xor-2 steps every lane of a quad holds the same value, so row_half_mirror / row_mirror stand in for xor 4 / xor 8.)
__device__ __forceinline__ float lane_xor1(float v) { return dpp_mov<0xB1>(v); }    // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_mov<0x4E>(v); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_hmirror(float v) { return dpp_mov<0x141>(v); } // row_half_mirror
__device__ __forceinline__ float lane_mirror(float v) { return dpp_mov<0x140>(v); }  // row_mirror
__device__ __forceinline__ float lane_xor16(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
// all-reduce over the 16 lanes of a DPP row (lanes 16k..16k+15)
__device__ __forceinline__ float row16_sum(float v) {
  v += lane_xor1(v);
  v += lane_xor2(v);
  v += lane_hmirror(v);
  v += lane_mirror(v);
  return v;
}
// all-reduce across the four 16-lane groups (same lane&15)
// max of three without the NaN-quieting v_max_f32 x,x canonicalisations hipcc puts in front of fmaxf chains
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// v_permlane16_swap(a, b) exchanges the odd rows of a with the even rows of b: with a = b = v it returns
// (v of the even row of my row pair, v of the odd row) in every lane -- both halves of the butterfly step at once, so a
// reduction needs no select.  v_permlane32_swap likewise returns (lower 32 lanes, upper 32 lanes).
__device__ __forceinline__ float groups_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float groups_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// IEEE-754-2019 maximum (NaN-propagating): lowers to v_maximum3_f32 on gfx950 and needs none of the `v_max_f32 x, x`
// quieting operations llvm.maxnum (fmaxf) gets in IEEE mode in front of values the compiler cannot prove canonical
// (MFMA results): a 16-score row maximum is 8 instructions instead of 23.
__device__ __forceinline__ float fmaximum(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float groups_maximum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaximum(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaximum(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) { return groups_sum(row16_sum(v)); }

__device__ __forceinline__ float wave_xor_max(float v, int m) { return fmaxf(v, __shfl_xor(v, m, kWave)); }
__device__ __forceinline__ float wave_xor_sum(float v, int m) { return v + __shfl_xor(v, m, kWave); }

// ---- write-through stores (sc1): data another kernel's workgroups on OTHER XCDs will read.  A plain store leaves the line
//      dirty in this XCD's L2 until the end-of-kernel write-back, which then serialises behind the kernel (measured: the q/k/V^T
//      kernel's 25 MB cost ~4 us after its last wave); a write-through store sends the bytes on while the kernel still runs.
//      Buffer addressing: wave-uniform base in a descriptor + a 32-bit byte offset per lane (the host checks < 4 GiB).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct WtBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ explicit WtBuf(const void* base)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000)) {}
  __device__ __forceinline__ void store16(unsigned byte_off, u16x8 v) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, byte_off, 0, 16);   // aux 16 = sc1
  }
  __device__ __forceinline__ void store16(unsigned byte_off, float4 v) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, byte_off, 0, 16);
  }
  __device__ __forceinline__ void store8(unsigned byte_off, u16x4 v) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsrc, byte_off, 0, 16);
  }
};

// ---- hand-off between workgroups of ONE launch (the merged q/k/v + trajectory kernels): bytes stored write-through (sc1) by one
//      workgroup are read by its sibling workgroups with sc1 loads -- served by the L2 / the memory side, never by a CU's own L1,
//      which no other CU's store refreshes.  Protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"): every storing wave
//      waits vmcnt(0), workgroup barrier, ONE lane adds to the counter; a consumer wave polls the counter with an sc1 load and
//      issues its sc1 loads of the bytes only after the poll matched.
struct ScBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ explicit ScBuf(const void* base)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000)) {}
  // 16 bytes at base + voff + soff (soff wave-uniform); aux 16 = sc1
  __device__ __forceinline__ u16x8 load16(unsigned voff, unsigned soff) const {
    return __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16));
  }
};
// one dword, sc1, waited for (the poll of a counter another workgroup adds to)
__device__ __forceinline__ unsigned ld_sc1_u32(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// every vector-memory operation of this wave has completed (stores acknowledged)
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- in-kernel phase stamps: diagnostic builds only (-DAXVS_STAMPS); the shipped library contains none ----
#ifdef AXVS_STAMPS
static __device__ unsigned long long g_stamps[64 * 64];   // per translation unit; g_stamps[slot * 64 + (workgroup % 8) * 8 + wave]
// Stamps stay in SGPRs until AXVS_STAMP_FLUSH at the end of the kernel: no VGPR cost where registers are tight
// (a first version that stored each stamp immediately pushed a 250-VGPR kernel into spills and mis-measured it).
#define AXVS_STAMP_DECL unsigned long long st_[24] = {}
#define AXVS_STAMP(slot)                                                                      \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[slot])::"memory");      \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
#define AXVS_STAMP_FLUSH_AT(base, n)                                                          \
  do {                                                                                        \
    if (blockIdx.x < 8 && (threadIdx.x & 63) == 0)                                            \
      for (int i_ = 0; i_ < (n); ++i_) ::axvs::g_stamps[((base) + i_) * 64 + (blockIdx.x & 7) * 8 + (threadIdx.x >> 6)] = st_[i_]; \
  } while (0)
#define AXVS_STAMP_FLUSH(n) AXVS_STAMP_FLUSH_AT(0, n)
#else
#define AXVS_STAMP_DECL
#define AXVS_STAMP(slot)
#define AXVS_STAMP_FLUSH(n)
#define AXVS_STAMP_FLUSH_AT(base, n)
#endif

// ---- per-workgroup wall times: diagnostic builds only (-DAXVS_STAMPS_WG; tools/r5/wg_times.py).  The 100 MHz real-time counter is the same on every
//      CU, so start / end of all workgroups of a launch can be laid side by side: how long before the slowest workgroup the others finish.
#ifdef AXVS_STAMPS_WG
static __device__ unsigned long long g_wg[2 * 1024 * 4];      // [kernel kind][workgroup][start, end, XCC id, CU id]
#define AXVS_WG_BEGIN unsigned long long wg_t0_ = __builtin_amdgcn_s_memrealtime()
#define AXVS_WG_END(kind)                                                                                         \
  do {                                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                                  \
      unsigned long long* q_ = ::axvs::g_wg + ((kind) * 1024 + blockIdx.x) * 4;                                   \
      q_[0] = wg_t0_; q_[1] = __builtin_amdgcn_s_memrealtime();                                                   \
      q_[2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) /* HW_REG_XCC_ID[3:0] */;               \
      q_[3] = __builtin_amdgcn_s_getreg((4 << 0) | (8 << 6) | (3 << 11)) /* HW_REG_HW_ID: CU_ID[11:8] */;         \
    }                                                                                                             \
  } while (0)
#else
#define AXVS_WG_BEGIN
#define AXVS_WG_END(kind)
#endif

// lgkmcnt is a 4-bit counter.  hipcc (ROCm 7.2) will happily leave 16 or more LDS/SMEM operations in flight before one
// `s_waitcnt lgkmcnt(0)`; when the LDS is busy enough that none of them has returned by the time the 16th issues, the
// counter wraps and the wait falls through early (observed on gfx950: intermittent stale reads in a fused epilogue with
// 15 ds_read + 1 s_load outstanding).  Call this between batches of LDS reads so no path ever has more than ~12 in flight;
// tools/check_lgkm.py scans the generated ISA for violations.
__device__ __forceinline__ void lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace axvs
