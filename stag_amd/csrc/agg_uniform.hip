// agg_uniform.hip — instantiations of agg_kernel for noise kind "uniform" (see agg_kernel.hpp).
#include "agg_kernel.hpp"

namespace stag {
template <>
hipError_t agg_launch<kUniform>(const AggArgs& a, bool vec, hipStream_t stream) {
  return agg_launch_impl<kUniform>(a, vec, stream);
}
}  // namespace stag
