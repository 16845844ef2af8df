// nn_kernels.hip -- translation unit of the network kernels.
#include "nn_kernels.hpp"
#include "nn_tower32.hpp"
#ifdef SC_EXP
#include "../../tools/experiments/nn_tower16.hpp"   // not part of the product tree
#endif
#include "tower_config.hpp"
#define K_T32N scnn::k_tower32<scnn::PrecBF16, 128, SC_T32_RS, SC_T32_TPI>
#define K_T32W scnn::k_tower32<scnn::PrecBF16, 256, SC_T32W_RS, SC_T32W_TPI, SC_T32W_AB>
#define K_T8N scnn::k_tower32<scnn::PrecFP8, 128, SC_T8_RS, SC_T8_TPI, SC_T8_AB>
#define K_T8W scnn::k_tower32<scnn::PrecFP8, 256, SC_T8W_RS, SC_T8W_TPI, SC_T8W_AB>

#include <stdlib.h>

#include <algorithm>

#include "launchers.hpp"

namespace scl {
size_t tower16_lds_bytes(int C) {
    size_t cp = (size_t)C + 16, hp = scnn::HEAD + 16, rp = (size_t)C + 4;
    size_t rs = std::max<size_t>(64 * rp * 4, 64 * hp * 2);
    return 100 * cp * 2 + rs + 1024 * 4 + 768 * 4 + 8 * 4 + 4 * 64 * 8;
}
size_t tower_lds_bytes(int C) {
#ifdef SC_EXP
    return std::max<size_t>(tower16_lds_bytes(C), (size_t)scnn::tower32_lds_bytes(C));
#else
    return (size_t)scnn::tower32_lds_bytes(C);
#endif
}
// Production build: the channel-major tower (nn_tower32.hpp), one instantiation per trunk width.  Experiment builds
// (-DSC_EXP, tools/build_exp.sh) also carry the pixel-major 16x16x32 kernel (nn_kernels.hpp) of each width for A/B runs
// (SC_TOWER_V=1 at engine creation picks its weight packing, and with it the kernel).
const char* nn_init() {
    hipError_t e = hipSuccess;
    const void* kn[4] = {reinterpret_cast<const void*>(&K_T32N), reinterpret_cast<const void*>(&K_T32W),
                         reinterpret_cast<const void*>(&K_T8N), reinterpret_cast<const void*>(&K_T8W)};
    for (int i = 0; i < 4; i++) {
        e = hipFuncSetAttribute(kn[i], hipFuncAttributeMaxDynamicSharedMemorySize, scnn::tower32_lds_bytes(i & 1 ? 256 : 128, i >= 2));
        if (e != hipSuccess) return hipGetErrorString(e);
    }
#ifdef SC_EXP
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scnn::k_tower<256, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)tower16_lds_bytes(256));
    if (e != hipSuccess) return hipGetErrorString(e);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scnn::k_tower<128, 12, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)tower16_lds_bytes(128));
    if (e != hipSuccess) return hipGetErrorString(e);
    if (getenv("SC_EXP_WAND")) {
        int v = (int)strtol(getenv("SC_EXP_WAND"), nullptr, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(scnn::g_exp_wand), &v, sizeof(int));
    }
#endif
    return nullptr;
}
bool tower_variant_available(int C, bool tower32) {
#ifdef SC_EXP
    (void)C;
    (void)tower32;
    return true;
#else
    (void)C;
    return tower32;
#endif
}
void tower(const scnn::TowerArgs& a, hipStream_t s) {
    if (a.n_pos <= 0) return;
    const scnn::TowerArgs& b = a;
    if (a.net.tower32 && a.net.fp8 && a.net.C == 128)
        hipLaunchKernelGGL(K_T8N, dim3(a.n_pos), dim3(256), scnn::tower32_lds_bytes(128, true), s, b);
    else if (a.net.tower32 && a.net.fp8)
        hipLaunchKernelGGL(K_T8W, dim3(a.n_pos), dim3(256), scnn::tower32_lds_bytes(256, true), s, b);
    else if (a.net.tower32 && a.net.C == 128)
        hipLaunchKernelGGL(K_T32N, dim3(a.n_pos), dim3(256), scnn::tower32_lds_bytes(128), s, b);
    else if (a.net.tower32)
        hipLaunchKernelGGL(K_T32W, dim3(a.n_pos), dim3(256), scnn::tower32_lds_bytes(256), s, b);
#ifdef SC_EXP
    else if (a.net.C == 256)
        hipLaunchKernelGGL((scnn::k_tower<256, 4, 1>), dim3(a.n_pos), dim3(256), tower16_lds_bytes(256), s, b);
    else
        hipLaunchKernelGGL((scnn::k_tower<128, 12, 3>), dim3(a.n_pos), dim3(256), tower16_lds_bytes(128), s, b);
#endif
}
void value_fc1(const scnn::Fc1Args& a, hipStream_t s) {
    if (a.n_pos <= 0 || a.ksplit < 64 || scnn::FC1_K % (a.ksplit * 32)) return;   // the kernel's LDS tile holds K / 64 columns
    hipLaunchKernelGGL(scnn::k_value_fc1, dim3((a.n_pos + 63) / 64, a.ksplit), dim3(256), 0, s, a);
}
void value_finish(const scnn::VfinArgs& a, hipStream_t s) {
    if (a.n_pos <= 0) return;
    hipLaunchKernelGGL(scnn::k_value_finish, dim3(a.n_pos), dim3(64), 0, s, a);
}
}  // namespace scl
