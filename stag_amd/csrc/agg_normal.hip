// agg_normal.hip — instantiations of agg_kernel for noise kind "normal" (see agg_kernel.hpp).
#include "agg_kernel.hpp"

namespace stag {
template <>
hipError_t agg_launch<kNormal>(const AggArgs& a, bool vec, hipStream_t stream) {
  return agg_launch_impl<kNormal>(a, vec, stream);
}
}  // namespace stag
