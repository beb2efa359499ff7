// Cross-lane pieces the bundle-adjustment kernels share (ba.hip: dense lineariser; ba_sparse.hip: grouped pair stage).
#pragma once
#include "mqs_common.h"

namespace mqs {
namespace wave {

__device__ __forceinline__ void swap32(double &a, double &b)
{
    // v_permlane32_swap: a[lanes 32..63] <-> b[lanes 0..31]
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto rlo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto rhi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(rhi[0], rlo[0]);
    b = __hiloint2double(rhi[1], rlo[1]);
}

__device__ __forceinline__ void swap16(double &a, double &b)
{
    // v_permlane16_swap: odd 16-lane rows of a <-> even rows of b
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto rlo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto rhi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(rhi[0], rlo[0]);
    b = __hiloint2double(rhi[1], rlo[1]);
}

// Lane exchange v[lane ^ STRIDE] for STRIDE in {8, 4, 2, 1} with DPP moves (VALU latency; no LDS
// crossbar round trip as with ds_bpermute, which sat on the reduction's critical path).
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, BANK_MASK, false);
}

template <int STRIDE>
__device__ __forceinline__ double xor_lane(double v)
{
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    if (STRIDE == 8) {                       // row_ror:8 within each row of 16 lanes
        lo = dpp_mov<0x128, 0xf>(lo, lo);
        hi = dpp_mov<0x128, 0xf>(hi, hi);
    } else if (STRIDE == 4) {                // banks {0,2} take lane+4 (row_shl:4), banks {1,3} lane-4 (row_shr:4)
        const unsigned l0 = lo, h0 = hi;
        lo = dpp_mov<0x104, 0x5>(l0, l0);
        lo = dpp_mov<0x114, 0xa>(lo, l0);
        hi = dpp_mov<0x104, 0x5>(h0, h0);
        hi = dpp_mov<0x114, 0xa>(hi, h0);
    } else if (STRIDE == 2) {                // quad_perm [2,3,0,1]
        lo = dpp_mov<0x4e, 0xf>(lo, lo);
        hi = dpp_mov<0x4e, 0xf>(hi, hi);
    } else {                                 // quad_perm [1,0,3,2]
        lo = dpp_mov<0xb1, 0xf>(lo, lo);
        hi = dpp_mov<0xb1, 0xf>(hi, hi);
    }
    return __hiloint2double(hi, lo);
}

// Transposed wavefront reduction of 32 values per lane: on return lane l holds the sum over
// all 64 lanes of v[l >> 1].  v is destroyed.
__device__ __forceinline__ double wave_reduce32(double (&v)[32], int lane)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) { swap32(v[i], v[i + 16]); v[i] += v[i + 16]; }   // lane bit 5 selects i (+16)
#pragma unroll
    for (int i = 0; i < 8; ++i) { swap16(v[i], v[i + 8]); v[i] += v[i + 8]; }      // lane bit 4 selects i (+8)
    const bool b1 = lane & 2;
    // Lane bits 3 and 2: the half that keeps entry i adds its partner's i, the other half its partner's i + 4 (i + 2).  The halves are
    // whole DPP banks (four lanes each: bit 3 = banks {2, 3}, bit 2 = banks {1, 3}), so the exchange AND the choice are two DPP moves
    // per 32-bit half with complementary bank masks -- the lanes a move does not write keep `old`, their own other entry -- and one
    // addition of the two results: 5 instructions per entry where select + exchange + add took 7 (9 at stride 4).  The same two
    // operands meet in every lane as before (own + partner's): the same bits.
#if !defined(MQS_WAVE_REDUCE_SELECTS)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned alo = __double2loint(v[i]), ahi = __double2hiint(v[i]), blo = __double2loint(v[i + 4]), bhi = __double2hiint(v[i + 4]);
        // lanes with bit 3 clear (banks 0, 1): the partner's entry i; the others keep their own entry i + 4
        const double p = __hiloint2double(dpp_mov<0x128, 0x3>(bhi, ahi), dpp_mov<0x128, 0x3>(blo, alo));
        // lanes with bit 3 set (banks 2, 3): the partner's entry i + 4; the others keep their own entry i
        const double q = __hiloint2double(dpp_mov<0x128, 0xc>(ahi, bhi), dpp_mov<0x128, 0xc>(alo, blo));
        v[i] = p + q;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned alo = __double2loint(v[i]), ahi = __double2hiint(v[i]), blo = __double2loint(v[i + 2]), bhi = __double2hiint(v[i + 2]);
        // bit 2 clear (banks 0, 2): entry i of lane + 4 (row_shl:4); bit 2 set (banks 1, 3): entry i + 2 of lane - 4 (row_shr:4)
        const double p = __hiloint2double(dpp_mov<0x104, 0x5>(bhi, ahi), dpp_mov<0x104, 0x5>(blo, alo));
        const double q = __hiloint2double(dpp_mov<0x114, 0xa>(ahi, bhi), dpp_mov<0x114, 0xa>(alo, blo));
        v[i] = p + q;
    }
#else
    const bool b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double send = b3 ? v[i] : v[i + 4], keep = b3 ? v[i + 4] : v[i];
        v[i] = keep + xor_lane<8>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double send = b2 ? v[i] : v[i + 2], keep = b2 ? v[i + 2] : v[i];
        v[i] = keep + xor_lane<4>(send);
    }
#endif
    {
        const double send = b1 ? v[0] : v[1], keep = b1 ? v[1] : v[0];
        v[0] = keep + xor_lane<2>(send);
    }
    return v[0] + xor_lane<1>(v[0]);
}

// Sums of x and of y over the 64 lanes, the same on every lane, with the additions of the xor butterfly (v += v[lane ^ 32],
// ^ 16, ^ 8, ^ 4, ^ 2, ^ 1: the same pairs, hence the same bits as a __shfl_xor loop) but without its twelve dependent
// ds_bpermute round trips per value: the two values share the first stage (v_permlane32_swap leaves x's pair sums in lanes
// 0..31 and y's in lanes 32..63), the row-of-16 exchange is a v_permlane16_swap of the value with itself (a + b is the pair sum in
// every lane), the rest DPP moves; the totals come back through v_readlane.
__device__ __forceinline__ void sum2(double x, double y, double &sx, double &sy)
{
    swap32(x, y);                              // lanes 0..31: (x, x[l + 32]);  lanes 32..63: (y[l - 32], y)
    double t = x + y;
    double a = t, b = t;
    swap16(a, b);                              // even rows: (t, t[l + 16]);  odd rows: (t[l - 16], t)
    t = a + b;
    t += xor_lane<8>(t);
    t += xor_lane<4>(t);
    t += xor_lane<2>(t);
    t += xor_lane<1>(t);
    const int lo = __double2loint(t), hi = __double2hiint(t);
    sx = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    sy = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
}
__device__ __forceinline__ double sum1(double v)
{
    double s, unused;
    sum2(v, 0.0, s, unused);
    return s;
}

}  // namespace wave
}  // namespace mqs
