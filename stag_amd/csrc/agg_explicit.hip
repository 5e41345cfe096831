// agg_explicit.hip — instantiations of agg_kernel for noise kind "explicit" (see agg_kernel.hpp).
#include "agg_kernel.hpp"

namespace stag {
template <>
hipError_t agg_launch<kExplicit>(const AggArgs& a, bool vec, hipStream_t stream) {
  return agg_launch_impl<kExplicit>(a, vec, stream);
}
}  // namespace stag
