// CTC prefix beam search on device -- placeholder until the kernel lands.
#include "kernels.h"
namespace chiron {
size_t beam_workspace_bytes(int B, int T, int beam) { return 16; }
int launch_beam(const BeamParams& p, hipStream_t stream) { return -1; }
}  // namespace chiron
