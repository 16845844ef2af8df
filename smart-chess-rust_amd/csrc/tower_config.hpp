// tower_config.hpp -- ring / tap-group geometries of the tower instantiations (shared by nn_kernels.hip and step_kernels.hip)
#pragma once
#ifndef SC_T32_RS
#define SC_T32_RS 12   // narrow trunk: 12-slot weight ring, all 9 taps of a conv unrolled (no tap-group loop: measured
#define SC_T32_TPI 9   // -4 % cycles, -1 % wall over groups of 3; experiment builds may override)
#endif

#ifndef SC_T32W_RS
#define SC_T32W_RS 8    // wide trunk: 8-slot ring, all 9 taps of a conv unrolled.  Round 1 measured RS 8/12 x TPI 1/3/9 within 0.5 % of each
#define SC_T32W_TPI 9   // other in the stand-alone tower; in the fused step kernel the same-box A/B of round 3 (tools/ab_r02.py, 10x256,
#endif                  // 256 games) gives TPI 9 +3.0 % over TPI 1 (RS 8, 9, 12 alike; TPI 3 +0.2 %; RS 4 -13 %)

#ifndef SC_T32W_AB
#define SC_T32W_AB 4   // image-fragment buffers of the wide trunk (SC_T32_AB, nn_tower32.hpp, for the narrow one)
#endif

// fp8 (e4m3) towers: k-steps of 64 (half as many, twice as long: the same bytes in flight need half the ring slots)
#ifndef SC_T8_RS
#define SC_T8_RS 6
#define SC_T8_TPI 3
#define SC_T8_AB 2   // image-fragment buffers: 2 instead of 3 measures +0.7 % at 512 games (configs[4]'s per-GPU share), -0.1 % at 256 (same-box A/B)
#endif
#ifndef SC_T8W_RS
#define SC_T8W_RS 6
#define SC_T8W_TPI 3
#define SC_T8W_AB 3
#endif
