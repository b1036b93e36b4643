// placeholder until the LDS-tiled kernels land: variant 2 forwards to variant 1
#include "bfd_internal.h"
void bfd_launch_stress_v2(const bfd_dev &d, hipStream_t s) { bfd_launch_stress_v1(d, s); }
void bfd_launch_velocity_v2(const bfd_dev &d, hipStream_t s) { bfd_launch_velocity_v1(d, s); }
