#include "common.h"
#include <stdarg.h>
#include <string.h>

namespace tr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace tr

extern "C" int tr_abi_version(void) { return TR_ABI_VERSION; }
extern "C" const char* tr_last_error(void) { return tr::g_err; }
extern "C" size_t tr_conv3x3_packed_bytes(int cin, int cout, int compute) {
    tr::PackGeom g = tr::pack_geom(cin, cout, compute);
    return (size_t)g.ncg * g.nchunks * 9 * g.cgw * 64;
}
extern "C" size_t tr_conv_packed_bytes(int cin, int cout, int compute, int ksize) {
    tr::PackGeom g = tr::pack_geom(cin, cout, compute);
    return (size_t)g.ncg * g.nchunks * ksize * ksize * g.cgw * 64;
}
