/*
 * bsw_fetch_kernel.hip — GPU-side target fetch from a device-resident 2-bit packed reference
 * (replaces the host's bns_get_seq + reversal for the extension targets so that
 * only the read crosses PCIe).  HBM-bound byte work: one wavefront per target, 16 bases per lane
 * per step, written as the kernels' 4-bit words (bsw_device.h).
 *
 * Coordinates are bwa's: x in [0, l_pac) is the forward strand, x in [l_pac, 2*l_pac) the reverse
 * complement: base(x) = 3 - pac[2*l_pac - 1 - x]  (bns_get_seq).  pac packs 4 bases per byte, first
 * base in the top two bits.  Left-extension targets are produced reversed (dir = -1), as
 * mem_chain2aln hands them to ksw_extend2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bsw_device.h"

namespace bsw {

__device__ __forceinline__ uint32_t pac_base(const uint8_t *__restrict__ pac, int64_t l)
{
    return (uint32_t)(pac[l >> 2] >> ((~l & 3) << 1)) & 3u;
}

__global__ __launch_bounds__(256) void bsw_fetch_targets(const uint8_t *__restrict__ pac, const int64_t l_pac,
                                                         const bsw_fetch_desc *__restrict__ desc, const uint32_t nd,
                                                         uint64_t *__restrict__ seq)
{
    const uint32_t di = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (di >= nd) return;
    const bsw_fetch_desc d = desc[di];
    const int nw = ((int)d.tlen + 15) >> 4;
    for (int w = lane; w < nw; w += 64) {
        uint64_t v = 0;
        const int i0 = w * 16, i1 = min(i0 + 16, (int)d.tlen);
        for (int i = i0; i < i1; ++i) {
            const int64_t x = d.x0 + (int64_t)d.dir * i;
            const uint32_t b = x < l_pac ? pac_base(pac, x) : 3u - pac_base(pac, (l_pac << 1) - 1 - x);
            v |= (uint64_t)b << ((i - i0) * 4);
        }
        seq[d.dst_word + w] = v;
    }
}

hipError_t launch_fetch(const uint8_t *pac, int64_t l_pac, const bsw_fetch_desc *desc, uint32_t nd, uint64_t *seq, hipStream_t s)
{
    if (nd == 0) return hipSuccess;
    hipLaunchKernelGGL(bsw_fetch_targets, dim3((nd + 3u) / 4u), dim3(256), 0, s, pac, l_pac, desc, nd, seq);
    return hipGetLastError();
}

}  // namespace bsw
