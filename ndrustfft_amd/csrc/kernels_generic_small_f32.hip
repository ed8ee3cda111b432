// kernels_generic_small_f32.hip -- one instantiation set of the generic lane kernel (see generic_kernel.h);
// split into four translation units so they compile in parallel.
#include "generic_kernel.h"
namespace ndfft {
template int launch_generic_class<float, false>(const GenArgs<float> &, int, size_t, hipStream_t);
}
