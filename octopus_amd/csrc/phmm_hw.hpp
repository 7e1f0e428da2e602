// gfx950 intrinsic layer used by phmm_kernels.hpp: DPP lane shifts, byte permute, packed-int16 VALU.
//
// The product build (hipcc --offload-arch=gfx950) uses the real builtins below. The unit tests can compile the
// SAME kernel source for the host with -DOCTPHMM_SIM, in which case tests/sim/hipsim.hpp provides lockstep
// emulations of exactly these functions (test infrastructure only; never part of liboct_phmm.so).
#pragma once
#include <stdint.h>

#if defined(OCTPHMM_SIM)
#include "hipsim.hpp"
#else
#include <hip/hip_runtime.h>

#define OCT_DEVICE __device__ __forceinline__
#define OCT_DEVICE_NOINLINE __device__ __noinline__
#define OCT_HD __host__ __device__ __forceinline__
#define OCT_KERNEL(name) __global__ void name
#define OCT_MAX_THREADS(n) __launch_bounds__(n)
#define OCT_DYN_SMEM(ptr) extern __shared__ __attribute__((aligned(16))) unsigned char ptr[]

namespace octphmm { namespace hw {

// lane i <- lane i-1 within each 16-lane row; first lane of a row keeps `fill`
OCT_DEVICE uint32_t dpp_row_shr1(uint32_t fill, uint32_t v)  { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x111, 0xf, 0xf, false); }
// lane i <- lane i+1 within each 16-lane row; last lane of a row keeps `fill`
OCT_DEVICE uint32_t dpp_row_shl1(uint32_t fill, uint32_t v)  { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x101, 0xf, 0xf, false); }
// same across the whole wave64 (gfx9 DPP wave_shr:1 / wave_shl:1)
// the same shifts with zero in the lane that has no source (bound_ctrl): `shift | lane-constant fill` then folds into ONE v_or_b32_dpp
OCT_DEVICE uint32_t dpp_row_shr1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
OCT_DEVICE uint32_t dpp_row_shl1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }
OCT_DEVICE uint32_t dpp_wave_shr1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
OCT_DEVICE uint32_t dpp_wave_shl1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
// rotation inside each 16-lane row: lane i <- lane i+1, the row's last lane <- its first (row_ror:15)
OCT_DEVICE uint32_t dpp_row_rol1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x12F, 0xf, 0xf, false); }
// rotations of the whole wave: lane i <- lane i-1 with lane 0 <- lane 63 (ror), lane i <- lane i+1 with lane 63 <- lane 0 (rol)
OCT_DEVICE uint32_t dpp_wave_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C, 0xf, 0xf, false); }
OCT_DEVICE uint32_t dpp_wave_rol1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x134, 0xf, 0xf, false); }
OCT_DEVICE uint32_t dpp_wave_shr1(uint32_t fill, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false); }
OCT_DEVICE uint32_t dpp_wave_shl1(uint32_t fill, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false); }

// v_perm_b32: result byte n = byte sel.byte[n] of the 8-byte value {hi, lo} (0-3 -> lo, 4-7 -> hi, 0x0c -> 0x00, >= 0x0d -> 0xff)
OCT_DEVICE uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// v_pk_mad_u16: per 16-bit half a * b + c (wrapping)
// v_lshl_or_b32: (a << n) | b in one VALU op
template <int N> OCT_DEVICE uint32_t lshl_or_impl(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(N), "v"(b));
    return r;
}
#define OCT_LSHL_OR_DEFINED 1
OCT_DEVICE uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

OCT_DEVICE void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
OCT_DEVICE uint32_t shfl_xor(uint32_t v, int mask) { return (uint32_t)__shfl_xor((int)v, mask, 64); }
OCT_DEVICE uint32_t shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
OCT_DEVICE uint32_t readfirstlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
OCT_DEVICE uint64_t ballot(bool p) { return __ballot(p); }
OCT_DEVICE uint32_t readlane(uint32_t v, uint32_t src_uniform) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_uniform); }
// wave-wide unsigned max, result in every lane: DPP row shifts + row broadcasts (no LDS round trips), then a scalar broadcast
OCT_DEVICE uint32_t wave_max_u32(uint32_t v)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8  -> lane 15 of each row holds the row max
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave max
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// wave-wide unsigned min: lanes without a DPP source read 0xffffffff
OCT_DEVICE uint32_t wave_min_u32(uint32_t v)
{
    auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// wave-wide sum, result in every lane: the same DPP ladder (row prefix sums, then row broadcasts); six VALU instructions and one v_readlane,
// no scalar-unit work - the k-mer mapper is bound by the CU's single scalar ALU, so its reductions stay on the vector side
OCT_DEVICE uint32_t wave_sum_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8  -> lane 15 of each row holds the row sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
OCT_DEVICE uint32_t atomic_add_lds_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
OCT_DEVICE unsigned long long atomic_cas_lds_u64(unsigned long long* p, unsigned long long expected, unsigned long long v) { return atomicCAS(p, expected, v); }   // ds_cmpst_rtn_b64
OCT_DEVICE void atomic_min_lds_u64(unsigned long long* p, unsigned long long v) { atomicMin(p, v); }                                                               // ds_min_u64
OCT_DEVICE uint32_t atomic_max_lds_u32(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
OCT_DEVICE void block_sync() { __syncthreads(); }
OCT_DEVICE int  atomic_min_i32(int32_t* p, int32_t v) { return atomicMin(p, v); }
OCT_DEVICE unsigned long long atomic_min_u64(unsigned long long* p, unsigned long long v) { return atomicMin(p, v); }
OCT_DEVICE uint32_t atomic_or_u32(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
OCT_DEVICE unsigned long long atomic_max_u64(unsigned long long* p, unsigned long long v) { return atomicMax(p, v); }
OCT_DEVICE unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
OCT_DEVICE uint32_t atomic_and_u32(uint32_t* p, uint32_t v) { return atomicAnd(p, v); }
OCT_DEVICE uint32_t atomic_min_u32(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
OCT_DEVICE unsigned long long atomic_cas_u64(unsigned long long* p, unsigned long long expected, unsigned long long v) { return atomicCAS(p, expected, v); }
// 64-bit mailboxes in LDS between the waves of a workgroup (k_dp_mw): value and tag travel in ONE store / ONE load, re-read until the tag is right
OCT_DEVICE void lds_store_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
OCT_DEVICE unsigned long long lds_load_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// loads that see what other CUs' atomics have written (device scope: past this CU's L1, which is not coherent), for look-before-you-swap in hash tables
OCT_DEVICE unsigned long long load_device_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
OCT_DEVICE uint32_t load_device_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
OCT_DEVICE void spin_pause() { __builtin_amdgcn_s_sleep(1); }
// a 16-byte store that will not be read again before it has left every cache (the DP's backpointer tiles: written once, fetched by the walk kernel a launch later): non-temporal
typedef uint32_t u32x4_native __attribute__((ext_vector_type(4)));
OCT_DEVICE void store_streaming_u4(uint4* p, uint4 v) { u32x4_native w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (u32x4_native*)p); }
// issue priority of this wave among the waves of its SIMD (0 lowest ... 3): a launch that heads a chain of dependent launches (traceback DP -> walk) beside one that does not
template <int P> OCT_DEVICE void wave_priority() { __builtin_amdgcn_s_setprio(P); }
OCT_DEVICE uint32_t thread_idx() { return threadIdx.x; }
OCT_DEVICE uint32_t block_idx() { return blockIdx.x; }
OCT_DEVICE uint32_t block_dim() { return blockDim.x; }
OCT_DEVICE uint32_t grid_dim() { return gridDim.x; }

}} // namespace octphmm::hw
#endif

namespace octphmm { namespace hw {

#if defined(OCT_LSHL_OR_DEFINED)
#define hw_lshl_or(a, n, b) octphmm::hw::lshl_or_impl<(n)>((a), (b))
#else
#define hw_lshl_or(a, n, b) ((((uint32_t)(a)) << (n)) | (uint32_t)(b))
#endif

typedef short          s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// packed 2 x int16 VALU (v_pk_add_u16, v_pk_add_i16 clamp, v_pk_min_i16, v_pk_min_u16, v_pk_lshlrev_b16, v_pk_mul_lo_u16, v_pk_sub_u16)
OCT_DEVICE uint32_t pk_add(uint32_t a, uint32_t b)     { return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b))); }
OCT_DEVICE uint32_t pk_sub(uint32_t a, uint32_t b)     { return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b))); }
OCT_DEVICE uint32_t pk_add_sat(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
OCT_DEVICE uint32_t pk_add_sat_u(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
OCT_DEVICE uint32_t pk_min_i(uint32_t a, uint32_t b)   { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
OCT_DEVICE uint32_t pk_min_u(uint32_t a, uint32_t b)   { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
OCT_DEVICE uint32_t pk_shl2(uint32_t a)                { const u16x2 two = {2, 2}; return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) << two)); }
OCT_DEVICE uint32_t pk_mul(uint32_t a, uint32_t b)     { return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b))); }

}} // namespace octphmm::hw
