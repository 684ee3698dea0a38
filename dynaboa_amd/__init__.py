"""MI355X-native DynaBOA adaptation hot path."""
import os as _os

# The ROCm runtime multiplexes a process's HIP streams onto 4 hardware queues by default.  One sequence with the reference's default
# term set runs its chain, the weight-gradient stream and the two pass streams of a level side by side (adapt_step.hip "par_passes");
# with 4 queues two of them share one and serialise (77.7 frames/s), with 8 they do not (89 - 92).  Every other configuration
# measures the same with 4 and 8 (profiles/r05_sessions.txt s23 / s24).  Read by the runtime when it initialises: effective when this
# package is imported before the process's first HIP call; an explicit setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
