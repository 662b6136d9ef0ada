// lr_flat.hip -- the kernels of FLAT scenes (<= 32 primitives: k_path_flat, the flat instantiations of k_resident) as a translation
// unit of their own, because they want one compiler switch the tree kernels do not: -mllvm -enable-post-misched=false.
// Their hot loop is a long, fully converged run of VALU instructions over scalar-loaded primitive rows; LLVM's post-RA scheduler
// reorders it for latencies that the six resident waves hide anyway and lengthens the wave's own serial path.  Measured on MI355X,
// interleaved (gpurun_out/r04n/ab.log): configs[1] 5912 -> 6007 Msamples/s (+1.6 %), config 3 9899 -> 9965 (+0.7 %) at 2048 spp; the tree
// kernels lose with the same switch (config 5 -0.9 %, config 4 +0.1 %) and stay in lumilly_hip.hip with the default scheduler.
// Same source (lr_path.h / lr_kernels.h), same arithmetic: the switch moves instructions, it does not change one.
#define LR_TEMPLATE_KERNELS_ONLY 1
#include "lr_path.h"

namespace lr {

template __global__ void k_path_flat<1u, 1>(DevScene, DevState, DevParams, const float4*);
template __global__ void k_path_flat<9u, 1>(DevScene, DevState, DevParams, const float4*);
template __global__ void k_path_flat<31u, 1>(DevScene, DevState, DevParams, const float4*);
template __global__ void k_path_flat<1u, 2>(DevScene, DevState, DevParams, const float4*);
template __global__ void k_path_flat<9u, 2>(DevScene, DevState, DevParams, const float4*);
template __global__ void k_path_flat<31u, 2>(DevScene, DevState, DevParams, const float4*);

template __global__ void k_resident<true, 1u, 512>(DevScene, DevState, DevParams, uint32_t, const float4*);
template __global__ void k_resident<true, 1u, 256>(DevScene, DevState, DevParams, uint32_t, const float4*);
template __global__ void k_resident<true, 31u, 512>(DevScene, DevState, DevParams, uint32_t, const float4*);
template __global__ void k_resident<true, 31u, 256>(DevScene, DevState, DevParams, uint32_t, const float4*);

}  // namespace lr
