// opental_amd/csrc/core.hip -- ABI version + error strings for libopental_hip.so.
#include "common.h"

extern "C" int otal_abi_version(void) { return OTAL_ABI_VERSION; }

extern "C" const char* otal_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case OTAL_E_NULL: return "null pointer argument";
        case OTAL_E_SHAPE: return "non-positive or inconsistent size";
        case OTAL_E_ODD_C: return "BoundaryMaxPooling needs an even channel count";
        case OTAL_E_BATCH: return "segments batch differs from feature batch";
        case OTAL_E_DTYPE: return "unsupported dtype code";
        case OTAL_E_LEVELS: return "bad level table";
        case OTAL_E_UNSUPPORTED: return "configuration not supported by this build";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

// ---- tuning / ablation switches -------------------------------------------------------------------------------------
// Named integer options (kernel-selection switches used by the A/B tests and the micro-benchmarks).  A switch takes its
// initial value from the environment variable of the same name, read ONCE at its first lookup: the launch path never
// calls getenv (it used to, ~10 times per convolution launch).  otal_set_option changes a switch at run time.
#include <cstdlib>
#include <cstring>
#include <mutex>
namespace {
struct Option { char name[48]; int value; };
constexpr int MAX_OPTIONS = 96;
Option g_options[MAX_OPTIONS];
int g_noptions = 0;
std::mutex g_options_mutex;
int* find_or_add(const char* name, int dflt, bool from_env) {
    for (int i = 0; i < g_noptions; ++i)
        if (!strcmp(g_options[i].name, name)) return &g_options[i].value;
    if (g_noptions == MAX_OPTIONS || strlen(name) >= sizeof(Option::name)) return nullptr;
    Option& o = g_options[g_noptions];
    strcpy(o.name, name);
    o.value = dflt;
    if (from_env)
        if (const char* e = getenv(name)) {          // present: its number; present but not a number (or empty): 1
            o.value = atoi(e);
            if (o.value == 0 && e[0] != '0') o.value = 1;
        }
    return &g_options[g_noptions++].value;
}
}  // namespace

int* otal_option_slot(const char* name, int dflt) {
    std::lock_guard<std::mutex> lock(g_options_mutex);
    static int sink = 0;
    int* p = find_or_add(name, dflt, true);
    if (!p) { sink = dflt; return &sink; }
    return p;
}

extern "C" int otal_set_option(const char* name, int value) {
    if (!name) return OTAL_E_NULL;
    std::lock_guard<std::mutex> lock(g_options_mutex);
    int* p = find_or_add(name, value, false);
    if (!p) return OTAL_E_UNSUPPORTED;
    *p = value;
    return 0;
}

extern "C" int otal_get_option(const char* name, int dflt) {
    if (!name) return dflt;
    return *otal_option_slot(name, dflt);
}

// ---- stream fork / join ---------------------------------------------------------------------------------------------
// A ring of events: hipStreamWaitEvent takes the event's state at the time of the call, so an event may be recorded again
// once its wait has been issued.
extern "C" int otal_stream_wait(void* waiter, void* signaler) {
    constexpr int RING = 64;
    static hipEvent_t ring[RING];
    static int made = 0, next = 0;
    static std::mutex m;
    if (waiter == signaler) return 0;
    std::lock_guard<std::mutex> lock(m);
    if (made < RING && next == made) {
        if (hipError_t e = hipEventCreateWithFlags(&ring[made], hipEventDisableTiming)) return (int)e;
        ++made;
    }
    hipEvent_t ev = ring[next];
    next = (next + 1) % RING;
    if (hipError_t e = hipEventRecord(ev, (hipStream_t)signaler)) return (int)e;
    if (hipError_t e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0)) return (int)e;
    return 0;
}
