"""MI355X-native PackNet-SAN depth-edge-refinement path (see README.md / DESIGN.md)."""
import os

# The training step keeps two HIP streams busy (main chain + weight-gradient stream) and RCCL adds its own.  ROCm maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that land on one queue serialise, and the
# weight-gradient overlap is lost (measured: 31 -> 37 ms/step as soon as a process group exists).  The runtime reads the
# variable when HIP initialises, so this only helps if the package is imported before the first GPU call; the entry
# points (bench.py, train_edges.py, infer_edges.py) also set it first thing.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
