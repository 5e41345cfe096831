// agg_bernoulli.hip — instantiations of agg_kernel for noise kind "bernoulli" (see agg_kernel.hpp).
#include "agg_kernel.hpp"

namespace stag {
template <>
hipError_t agg_launch<kBernoulli>(const AggArgs& a, bool vec, hipStream_t stream) {
  return agg_launch_impl<kBernoulli>(a, vec, stream);
}
}  // namespace stag
