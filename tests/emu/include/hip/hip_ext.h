// CPU emulator stand-in for <hip/hip_ext.h>: events are no-ops in the single-threaded emulator, so
// the event-carrying launch is the ordinary launch.
#pragma once
#include <hip/hip_runtime.h>
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, start_ev, stop_ev, flags, ...) \
  hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
