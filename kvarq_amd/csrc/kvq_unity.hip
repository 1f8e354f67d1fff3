// single translation unit of libkvarq_hip.so (kernels and their launch sites
// must share one HIP module)
#include "kernels_general.hip"
#include "kernels_seeded.hip"
#include "kernels_bp.hip"
#include "kvq_launch.hip"
#include "kernels_results.hip"
#include "synth.hip"
#include "kvq_runtime.hip"
#include "kvq_findseqs.hip"
#include "kvq_dist.hip"
