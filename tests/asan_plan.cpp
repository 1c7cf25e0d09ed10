// Sanitizer run of the C++ batch manager's host logic (validation, chunk layout, class counts, plan replay): the host-side .hip files
// (bsw_ctx / bsw_batch / bsw_scalar / bsw_wire / bsw_f4) are compiled host-only with -fsanitize=address,undefined and linked against stubs for the kernel launchers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bwa_sw_mi355.h"
#include "../bwa-mem-sw_amd/csrc/bsw_device.h"
#include "../bwa-mem-sw_amd/csrc/bsw_stage.h"

namespace bsw {
static const int kW[] = {1, 2, 3, 4, 8, 16};
int wave_class_count() { return 6; }
int wave_class_cols(int c) { return kW[c] * 64; }
static const int kLb[] = {8, 8, 8, 16}, kLc[] = {72, 136, 232, 136};
int lane_class_count() { return 4; }
int lane_class_cols(int c) { return kLc[c]; }
int lane_class_bits(int c) { return kLb[c]; }
hipError_t launch_wave(int, int, const bsw_dparams &, const uint64_t *, const bsw_dtask *, const uint32_t *, uint32_t, const uint32_t *, uint32_t *, bsw_result *, hipStream_t) { return hipSuccess; }
bool lane_class_signals_tail(int) { return false; }
hipError_t launch_wait_count(const uint32_t *, uint32_t, uint32_t *, hipStream_t) { return hipSuccess; }
hipError_t launch_lane(int, int, const bsw_dparams &, int, const uint64_t *, const bsw_dtask *, const uint32_t *, uint32_t, bsw_result *, hipStream_t, uint32_t *, uint32_t *, const bsw_fin *) { return hipSuccess; }
bool lane_class_finishes(int, const bsw_dparams &, int) { return true; }
hipError_t launch_finalize(const bsw_dparams &, const bsw_dtask *, const uint32_t *, uint32_t, bsw_result *, uint32_t *, uint32_t *, bsw_pair *, hipStream_t) { return hipSuccess; }
hipError_t launch_pairs_from_results(const uint32_t *, uint32_t, const uint32_t *, const bsw_result *, bsw_pair *, hipStream_t) { return hipSuccess; }
hipError_t launch_pack(const uint8_t *, const bsw_dtask *, const bsw_rawoff *, uint32_t, uint32_t, int, const uint8_t *, int64_t, const bsw_refx *, uint64_t *, uint8_t *, hipStream_t) { return hipSuccess; }
hipError_t launch_rebase(bsw_dtask *, const bsw_rawoff *, uint32_t, const bsw_rebase &, hipStream_t) { return hipSuccess; }
hipError_t launch_wire_pack(const uint32_t *, const bsw_dtask *, const bsw_wireoff *, uint32_t, uint64_t *, hipStream_t) { return hipSuccess; }
hipError_t launch_wire_results(const bsw_result *, const bsw_wireoff *, uint32_t, uint32_t *, size_t, hipStream_t) { return hipSuccess; }
hipError_t launch_bin(const bsw_binparams &, const uint64_t *, const uint8_t *, const bsw_dtask *, uint32_t, uint32_t *, uint64_t *, uint32_t *, hipStream_t) { return hipSuccess; }
int align_class_count() { return 6; }
int align_class_of(int qlen, int byte_mode) { return qlen <= 128 ? (byte_mode ? 0 : 3) : (qlen <= 160 ? (byte_mode ? 1 : 4) : (qlen <= 256 ? (byte_mode ? 2 : 5) : -1)); }
hipError_t launch_align(int, const bsw_dparams &, const uint64_t *, const bsw_adtask *, const uint32_t *, uint32_t, unsigned long long *, bsw_kswr *, hipStream_t) { return hipSuccess; }
int global_class_count() { return 5; }
int global_class_cols(int c) { return (1 << c) * 64; }
hipError_t launch_global(int, const bsw_dparams &, const uint64_t *, const bsw_gdtask *, const uint32_t *, uint32_t, uint8_t *, uint32_t *, int, bsw_gresult *, hipStream_t) { return hipSuccess; }
}

int main()
{
    bsw_params p; bsw_default_params(&p);
    bsw_synth_spec sp; memset(&sp, 0, sizeof(sp));
    sp.seed = 3; sp.read_len = 250; sp.seed_len_min = 19; sp.seed_len_max = 80; sp.sub_rate = 0.03; sp.indel_rate = 0.01;
    sp.n_rate = 0.003; sp.junk_frac = 0.1; sp.a = 1; sp.w = 100; sp.o = 6; sp.e = 1;
    for (size_t n : {size_t(0), size_t(1), size_t(63), size_t(5000), size_t(20000)}) {
        std::vector<uint8_t> arena(bsw_synth_arena_bound(&sp, n));
        std::vector<bsw_task> tasks(n ? n : 1);
        if (bsw_synth_generate(&sp, n, tasks.data(), arena.data(), arena.size()) < 0) return 1;
        for (int kernel = 0; kernel < 3; ++kernel)
            for (int threads : {1, 5}) {
                std::vector<uint32_t> order(4 * n + 16), seg(BSW_PLAN_SEGS + 1);
                if (bsw_plan_batch(&p, tasks.data(), n, kernel, threads, order.data(), seg.data()) < 0) return 2;
                if (seg[BSW_PLAN_SEGS] > 4 * n + 15) return 3;
            }
    }
    uint8_t b[77]; uint64_t w[5];
    for (int i = 0; i < 77; ++i) b[i] = (uint8_t)(i * 37);
    for (int len = 0; len <= 77; ++len) bsw_pack_bases(b, len, w);
    puts("asan_plan ok");
    return 0;
}
