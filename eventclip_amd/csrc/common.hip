// Error reporting, version and device info for libeventclip_hip.so.
#include "common.h"

#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

namespace ec {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {
std::mutex g_attr_mu;
std::vector<std::pair<const void *, int>> g_attr_done;   // (kernel, device) pairs already raised
int g_cus[64] = {0};
}  // namespace

int ensure_dynamic_lds(const void *kern, int bytes)
{
    int dev = 0;
    EC_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    for (const auto &d : g_attr_done)
        if (d.first == kern && d.second == dev) return EC_OK;
    EC_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_attr_done.emplace_back(kern, dev);
    return EC_OK;
}

int cu_count()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (!g_cus[dev] &&
        hipDeviceGetAttribute(&g_cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        g_cus[dev] = 0;
    return g_cus[dev];
}

namespace {
struct ProfRec {
    int cls;
    double flops, bytes;
    hipEvent_t e0, e1;
};
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;   // events are recycled across begin/end cycles
size_t g_pool_used = 0;
std::mutex g_prof_mu;
constexpr size_t PROF_MAX = 1 << 16;
const char *const kProfNames[PROF_NCLASS] = {
    "events_to_frames_kernel", "preprocess_kernel", "patchify_kernel", "gemm_kernel<STORE16>",
    "gemm_kernel<GELU16>", "gemm_kernel<RESID32>", "gemm_kernel<STORE32>", "layernorm_kernel",
    "attention_kernel", "embed_kernel", "classify_kernel", "adapter_kernels",
    "gemm_kernel<STORE32 K-batches>", "attention_dq_kernel + attention_dkv_kernel", "ln_bwd_kernel",
    "transpose_kernel", "colsum_kernel + reduce_kernel", "sgemm_kernel", "adam_kernel + unscale_check_kernel",
    "pack_weight_kernel"};

hipEvent_t take_event()
{
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_used++];
}
}  // namespace

ProfScope::ProfScope(int cls, hipStream_t s, double flops, double bytes) : slot(-1), stream(s)
{
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_recs.size() >= PROF_MAX) return;
    ProfRec r{cls, flops, bytes, take_event(), take_event()};
    if (!r.e0 || !r.e1) return;
    (void)hipEventRecord(r.e0, s);
    slot = (int)g_recs.size();
    g_recs.push_back(r);
}

ProfScope::~ProfScope()
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_recs[slot].e1, stream);
}

}  // namespace ec

extern "C" {

EC_API int ec_profile_begin(void)
{
    std::lock_guard<std::mutex> lk(ec::g_prof_mu);
    ec::g_recs.clear();
    ec::g_pool_used = 0;
    ec::g_prof_on = true;
    return EC_OK;
}

EC_API int ec_profile_end(ec_profile_entry *out, int cap, int *n_out)
{
    std::lock_guard<std::mutex> lk(ec::g_prof_mu);
    ec::g_prof_on = false;
    ec_profile_entry acc[ec::PROF_NCLASS];
    for (int c = 0; c < ec::PROF_NCLASS; c++) {
        memset(&acc[c], 0, sizeof(acc[c]));
        snprintf(acc[c].name, sizeof(acc[c].name), "%s", ec::kProfNames[c]);
    }
    for (auto &r : ec::g_recs) {
        EC_CHECK_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        EC_CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        acc[r.cls].launches += 1;
        acc[r.cls].total_ms += ms;
        acc[r.cls].flops += r.flops;
        acc[r.cls].bytes += r.bytes;
    }
    int n = 0;
    for (int c = 0; c < ec::PROF_NCLASS; c++)
        if (acc[c].launches > 0 && out && n < cap) out[n++] = acc[c];
    if (n_out) *n_out = n;
    ec::g_recs.clear();
    return EC_OK;
}


EC_API const char *ec_last_error(void) { return ec::err_buf(); }

EC_API int ec_version(void) { return EC_ABI_VERSION; }

EC_API int ec_abi_check(int header_version, size_t gemm_args_bytes, size_t block_weights_bytes, size_t vit_weights_bytes,
                        size_t text_weights_bytes, size_t events_params_bytes, size_t adapter_weights_bytes)
{
    if (header_version != EC_ABI_VERSION)
        return ec::fail(EC_ERR_INVALID, "ec_abi_check: the caller was built against ABI %d, this library is ABI %d: rebuild the "
                                        "caller against include/eventclip_hip.h of this library", header_version, EC_ABI_VERSION);
    const struct { const char *name; size_t got, want; } s[] = {
        {"ec_gemm_args", gemm_args_bytes, sizeof(ec_gemm_args)},
        {"ec_block_weights", block_weights_bytes, sizeof(ec_block_weights)},
        {"ec_vit_weights", vit_weights_bytes, sizeof(ec_vit_weights)},
        {"ec_text_weights", text_weights_bytes, sizeof(ec_text_weights)},
        {"ec_events_params", events_params_bytes, sizeof(ec_events_params)},
        {"ec_adapter_weights", adapter_weights_bytes, sizeof(ec_adapter_weights)},
    };
    for (const auto &e : s)
        if (e.got != e.want)
            return ec::fail(EC_ERR_INVALID, "ec_abi_check: sizeof(%s) is %zu in the caller and %zu in this library (ABI %d): "
                                            "the caller's struct definitions are not this library's", e.name, e.got, e.want,
                            EC_ABI_VERSION);
    return EC_OK;
}

EC_API int ec_device_info(int *cu_count, char *name, int name_len)
{
    int dev = 0;
    EC_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EC_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return EC_OK;
}

}  // extern "C"
