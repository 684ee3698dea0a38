// CPU emulator stand-in for <hip/hip_ext.h>: the event-carrying launch records its start / stop events on the launch's stream around
// the kernel (no-ops unless the emulator's lazy stream mode is on, where a stop event is what another stream's wait orders against).
#pragma once
#include <hip/hip_runtime.h>
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, start_ev, stop_ev, flags, ...) \
  do {                                                                                           \
    hipEvent_t emu_e0_ = (start_ev), emu_e1_ = (stop_ev);                                        \
    if (emu_e0_) (void)hipEventRecord(emu_e0_, (stream));                                        \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                         \
    if (emu_e1_) (void)hipEventRecord(emu_e1_, (stream));                                        \
  } while (0)
