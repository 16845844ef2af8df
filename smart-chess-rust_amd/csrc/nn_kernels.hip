// nn_kernels.hip -- translation unit of the network kernels.
#include "nn_kernels.hpp"
#include "nn_tower32.hpp"
#ifndef SC_T32_RS
#define SC_T32_RS 12   // narrow trunk: 12-slot weight ring, all 9 taps of a conv unrolled (no tap-group loop: measured
#define SC_T32_TPI 9   // -4 % cycles, -1 % wall over groups of 3; experiment builds may override)
#endif

#include <stdlib.h>

#include <algorithm>

#include "launchers.hpp"

namespace scl {
size_t tower16_lds_bytes(int C);
size_t tower_lds_bytes(int C) {
    // both tower variants are sized; the larger one is what the callers reserve
    return std::max<size_t>(tower16_lds_bytes(C), (size_t)scnn::tower32_lds_bytes(C));
}
size_t tower16_lds_bytes(int C) {
    size_t cp = (size_t)C + 16, hp = scnn::HEAD + 16, rp = (size_t)C + 4;
    size_t rs = std::max<size_t>(64 * rp * 4, 64 * hp * 2);
    return 100 * cp * 2 + rs + 1024 * 4 + 768 * 4 + 8 * 4 + 4 * 64 * 8;
}
const char* nn_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scnn::k_tower<256, 4, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)tower_lds_bytes(256));
    if (e != hipSuccess) return hipGetErrorString(e);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scnn::k_tower<128, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)tower_lds_bytes(128));
    if (e != hipSuccess) return hipGetErrorString(e);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scnn::k_tower<128, 12, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)tower_lds_bytes(128));
    if (e != hipSuccess) return hipGetErrorString(e);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scnn::k_tower32<256, 8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            scnn::tower32_lds_bytes(256));
    if (e != hipSuccess) return hipGetErrorString(e);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scnn::k_tower32<128, 8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            scnn::tower32_lds_bytes(128));
    if (e != hipSuccess) return hipGetErrorString(e);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scnn::k_tower32<128, SC_T32_RS, SC_T32_TPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            scnn::tower32_lds_bytes(128));
    if (e != hipSuccess) return hipGetErrorString(e);
#ifdef SC_EXP
    if (getenv("SC_EXP_WAND")) {
        int v = (int)strtol(getenv("SC_EXP_WAND"), nullptr, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(scnn::g_exp_wand), &v, sizeof(int));
    }
#endif
    return nullptr;
}
void tower(const scnn::TowerArgs& a, hipStream_t s) {
    if (a.n_pos <= 0) return;
    static const int ring = getenv("SC_TOWER_RING") ? atoi(getenv("SC_TOWER_RING")) : 12;     // experiment knobs
    static const int stagger = getenv("SC_TOWER_STAGGER") ? atoi(getenv("SC_TOWER_STAGGER")) : 0;
    scnn::TowerArgs b = a;
    b.stagger = stagger;
    static const int delay = getenv("SC_TOWER_DELAY") ? atoi(getenv("SC_TOWER_DELAY")) : 0;
    b.delay = delay;
    if (a.net.tower32) {
        if (a.net.C == 256)
            hipLaunchKernelGGL((scnn::k_tower32<256, 8, 1>), dim3(a.n_pos), dim3(256), scnn::tower32_lds_bytes(256), s, b);
        else if (ring == 12)
            hipLaunchKernelGGL((scnn::k_tower32<128, SC_T32_RS, SC_T32_TPI>), dim3(a.n_pos), dim3(256), scnn::tower32_lds_bytes(128), s, b);
        else
            hipLaunchKernelGGL((scnn::k_tower32<128, 8, 1>), dim3(a.n_pos), dim3(256), scnn::tower32_lds_bytes(128), s, b);
    } else if (a.net.C == 256)
        hipLaunchKernelGGL((scnn::k_tower<256, 4, 1>), dim3(a.n_pos), dim3(256), tower_lds_bytes(256), s, b);
    else if (ring == 12)
        hipLaunchKernelGGL((scnn::k_tower<128, 12, 3>), dim3(a.n_pos), dim3(256), tower_lds_bytes(128), s, b);
    else
        hipLaunchKernelGGL((scnn::k_tower<128, 4, 1>), dim3(a.n_pos), dim3(256), tower_lds_bytes(128), s, b);
}
void value_fc1(const scnn::Fc1Args& a, hipStream_t s) {
    if (a.n_pos <= 0) return;
    hipLaunchKernelGGL(scnn::k_value_fc1, dim3((a.n_pos + 63) / 64, a.ksplit), dim3(256), 0, s, a);
}
void value_finish(const scnn::VfinArgs& a, hipStream_t s) {
    if (a.n_pos <= 0) return;
    hipLaunchKernelGGL(scnn::k_value_finish, dim3(a.n_pos), dim3(64), 0, s, a);
}
}  // namespace scl
