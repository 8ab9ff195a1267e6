#include "common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

namespace tr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int env_choice(const char* name, const char* one, const char* two) {
    const char* v = getenv(name);
    if (!v) return 0;
    return !strcmp(v, one) ? 1 : (!strcmp(v, two) ? 2 : 0);
}
Tune& tune() {
    static Tune t = [] {
        Tune x{};
        x.conv_impl = env_choice("TRAINNER_CONV_IMPL", "reg", "ring");
        x.ring_cfg = getenv("TRAINNER_RING_CFG") ? atoi(getenv("TRAINNER_RING_CFG")) : -1;
        x.ring_generic = getenv("TRAINNER_RING_GENERIC") ? 1 : 0;
        x.wgrad_impl = env_choice("TRAINNER_WGRAD_IMPL", "reg", "ring");
        x.wgrad_multi = (getenv("TRAINNER_WGRAD_MULTI") && !strcmp(getenv("TRAINNER_WGRAD_MULTI"), "0")) ? 0 : 1;
        x.wgrad_reduce = (getenv("TRAINNER_WGRAD_REDUCE") && !strcmp(getenv("TRAINNER_WGRAD_REDUCE"), "scalar")) ? 1 : 0;
        x.dgrad_image_ref = getenv("TRAINNER_DGRAD_IMAGE_REF") ? 1 : 0;
        x.linear_impl = (getenv("TRAINNER_LINEAR_IMPL") && !strcmp(getenv("TRAINNER_LINEAR_IMPL"), "reg")) ? 1 : 0;
        return x;
    }();
    return t;
}
}  // namespace tr

namespace {
int* tune_slot(const char* key) {
    tr::Tune& t = tr::tune();
    struct { const char* k; int* p; } tab[] = {{"conv_impl", &t.conv_impl}, {"ring_cfg", &t.ring_cfg}, {"ring_generic", &t.ring_generic},
                                              {"wgrad_impl", &t.wgrad_impl}, {"wgrad_multi", &t.wgrad_multi}, {"wgrad_reduce", &t.wgrad_reduce}, {"dgrad_image_ref", &t.dgrad_image_ref},
                                              {"linear_impl", &t.linear_impl}, {"linear_epilogue", &t.linear_epilogue},
                                              {"last_conv_family", &t.last_conv_family}, {"last_conv_instance", &t.last_conv_instance},
                                              {"ring_launches", &t.ring_launches}};
    for (auto& e : tab)
        if (key && !strcmp(key, e.k)) return e.p;
    return nullptr;
}
}  // namespace

extern "C" int tr_tune_set(const char* key, long long value) {
    if (key && !strcmp(key, "stamps")) { tr::tune().stamps = value; return 0; }
    int* p = tune_slot(key);
    TR_REQUIRE(p != nullptr, TR_EINVAL, "tr_tune_set: unknown key '%s'", key ? key : "(null)");
    *p = (int)value;
    return 0;
}
extern "C" long long tr_tune_get(const char* key) {
    if (key && !strcmp(key, "stamps")) return tr::tune().stamps;
    int* p = tune_slot(key);
    if (!p) { tr::set_error("tr_tune_get: unknown key '%s'", key ? key : "(null)"); return -1; }
    return *p;
}

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
