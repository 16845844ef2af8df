// tower_config.hpp -- ring / tap-group geometries of the tower instantiations (shared by nn_kernels.hip and step_kernels.hip)
#pragma once
#ifndef SC_T32_RS
#define SC_T32_RS 12   // narrow trunk: 12-slot weight ring, all 9 taps of a conv unrolled (no tap-group loop: measured
#define SC_T32_TPI 9   // -4 % cycles, -1 % wall over groups of 3; experiment builds may override)
#endif

#ifndef SC_T32W_RS
#define SC_T32W_RS 8    // wide trunk: 8-slot ring, one tap per loop iteration (RS 8/12 x TPI 1/3/9 all measured
#define SC_T32W_TPI 1   // within 0.5 % of each other)
#endif

// fp8 (e4m3) towers: k-steps of 64 (half as many, twice as long: the same bytes in flight need half the ring slots)
#ifndef SC_T8_RS
#define SC_T8_RS 6
#define SC_T8_TPI 3
#define SC_T8_AB 3
#endif
#ifndef SC_T8W_RS
#define SC_T8W_RS 6
#define SC_T8W_TPI 3
#define SC_T8W_AB 3
#endif
