// agg_none.hip — instantiations of agg_kernel for noise kind "none" (see agg_kernel.hpp).
#include "agg_kernel.hpp"

namespace stag {
template <>
hipError_t agg_launch<kNone>(const AggArgs& a, bool vec, hipStream_t stream) {
  return agg_launch_impl<kNone>(a, vec, stream);
}
}  // namespace stag
