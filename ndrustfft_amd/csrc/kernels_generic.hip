// kernels_generic.hip -- host-side pieces of the generic lane kernel: LDS sizing and the dispatch between the
// two register classes (generic_kernel.h: BIG = radices 11/12/13/16 present).
#include "generic_kernel.h"

namespace ndfft {

size_t generic_lds_bytes(int lpb, int pitch, size_t csize, int nbuf) {
    return kGenHeaderBytes + (size_t)nbuf * (size_t)lpb * (size_t)pitch * csize;
}
int generic_z_len(int len) { return len + (len >> 3) + 1; }
size_t generic_max_len(size_t csize) {
    size_t len = 1;
    while (generic_lds_bytes(1, generic_z_len((int)(len + 1)) | 1, csize) <= 160 * 1024) ++len;
    return len;
}

extern template int launch_generic_class<float, false>(const GenArgs<float> &, int, size_t, hipStream_t);
extern template int launch_generic_class<float, true>(const GenArgs<float> &, int, size_t, hipStream_t);
extern template int launch_generic_class<double, false>(const GenArgs<double> &, int, size_t, hipStream_t);
extern template int launch_generic_class<double, true>(const GenArgs<double> &, int, size_t, hipStream_t);

bool generic_needs_big(const int32_t *radix, int npass, const int32_t *radixM, int npassM) {
    for (int i = 0; i < npass; ++i) if (radix[i] > 10) return true;
    for (int i = 0; i < npassM; ++i) if (radixM[i] > 10) return true;
    return false;
}

template <typename T> int launch_generic(const GenArgs<T> &a, int threads, size_t lds_bytes, hipStream_t s) {
    if (generic_needs_big(a.radix, a.npass, a.radixM, a.blue ? a.npassM : 0)) return launch_generic_class<T, true>(a, threads, lds_bytes, s);
    return launch_generic_class<T, false>(a, threads, lds_bytes, s);
}
template int launch_generic<float>(const GenArgs<float> &, int, size_t, hipStream_t);
template int launch_generic<double>(const GenArgs<double> &, int, size_t, hipStream_t);

}  // namespace ndfft
