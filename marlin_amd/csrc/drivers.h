// Internal host-side drivers shared between translation units of libmarlin_hip.so.
#pragma once
#include "context.h"

namespace mh {
// NTT over Fr on device buffers (d_in may equal d_out); see ntt.cuh.
int ntt_device(Context& c, const void* d_in, void* d_out, uint32_t log_n, int inverse);
// the same for an input of in_len <= 2^log_n elements that stands for a zero-padded vector (no padded copy needed)
int ntt_device_len(Context& c, const void* d_in, uint64_t in_len, void* d_out, uint32_t log_n, int inverse);
// MSM: d_bases = G1Affine[n] (Montgomery), d_scalars = Fr[n]; out = Jacobian X||Y||Z (18 u64, Montgomery).
int msm_device(Context& c, const void* d_bases, const void* d_scalars, int is_mont, size_t n, uint64_t* out_xyz);
// a batch of independent MSMs through one launch sequence; out_xyz: njobs x 18 limbs
// shard = {rank, world} shards fixed-base groups by bucket range; partial[j] = 1 then marks out_xyz[j] as this rank's share
int msm_batch_device(Context& c, int njobs, const void* const* d_bases, const void* const* d_scalars, const size_t* ns, int is_mont,
                     uint64_t* out_xyz, const int* shard = nullptr, uint8_t* partial = nullptr);
// MSMs over strided selections (base first[j] + i * stride) of one tabled base set; fixed-base path only
int msm_batch_strided_device(Context& c, const BaseSet& bs, int njobs, const size_t* first, size_t stride, const void* const* d_scalars,
                             const size_t* ns, int is_mont, uint64_t* out_xyz);
// fixed-base window table for a base set (msm_fb.cuh); window_bits = 0 picks a width from the set's size
int bases_precompute(Context& c, BaseSet& bs, uint32_t window_bits);
// twiddle table (tw[2^(l-1) + e] = omega_{2^l}^e) covering at least log_n levels
int ensure_twiddles_public(Context& c, uint32_t log_n);
}  // namespace mh
