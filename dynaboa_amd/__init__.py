"""MI355X-native DynaBOA adaptation hot path.

Runtime note: the ROCm runtime multiplexes a process's HIP streams onto 4 hardware queues by default.  With the reference's default
term set a level runs its chain, the weight-gradient stream and the two pass streams side by side (adapt_step.hip "par_passes"); with
4 queues two of them share one, with GPU_MAX_HW_QUEUES=8 in the environment (before the first HIP call) they do not: one sequence
78 -> 89 adapted frames/s, 5 sequences 181 -> 198, 32 sequences 381 -> 385, dynamic loop at 32: 124 -> 132.  Not set here: the
frame-loss configurations measure the same with 4 and 8 (profiles/r05_sessions.txt s23 - s26, s32)."""
