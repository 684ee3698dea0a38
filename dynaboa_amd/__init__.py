"""MI355X-native DynaBOA adaptation hot path.

Runtime note: the ROCm runtime multiplexes a process's HIP streams onto 4 hardware queues by default.  ONE sequence with the reference's
default term set runs its chain, the weight-gradient stream and the two pass streams of a level side by side (adapt_step.hip
"par_passes"); with 4 queues two of them share one (78 adapted frames/s), with GPU_MAX_HW_QUEUES=8 in the environment they do not
(89 - 92).  Not set here: other configurations measure the same or - uploads inside the step - lower with 8
(profiles/r05_sessions.txt s23 - s26)."""
