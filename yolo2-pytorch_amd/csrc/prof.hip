// prof.hip — measurement hooks of libyolo2_hip.so (include/yolo2_hip.h: y2_prof_enable / y2_prof_count / y2_prof_get).
//
// bench.py's roofline legs need "that kernel's average launch duration, measured live with HIP events on the stream the kernel
// is launched on": while recording is on, Y2_LAUNCH (common.h) brackets every kernel launch with an event pair and stores
// (kernel name, executed multiply-add FLOPs, events).  Recording is a tooling mode: it is off by default, must not be used
// during hipGraph capture, and serialises nothing by itself (events are only synchronised when a record is read).
#include <string.h>
#include <vector>

#include "common.h"

int y2_prof_on = 0;

namespace {
struct Rec {
    const char* name;
    double flops;
    int tag;
    hipEvent_t e0, e1;
};
int cur_tag = 0;
std::vector<Rec>& recs() { static std::vector<Rec> r; return r; }
std::vector<hipEvent_t>& pool() { static std::vector<hipEvent_t> p; return p; }
size_t pool_used = 0;
bool open_rec = false;

hipEvent_t take_event() {
    auto& p = pool();
    if (pool_used == p.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        p.push_back(e);
    }
    return p[pool_used++];
}
}  // namespace

void y2_prof_begin(const char* name, hipStream_t s, double flops) {
    if (recs().size() >= (1u << 20)) { open_rec = false; return; }
    Rec r;
    r.name = name; r.flops = flops; r.tag = cur_tag; r.e0 = take_event(); r.e1 = take_event();
    if (r.e0 == nullptr || r.e1 == nullptr) { open_rec = false; return; }
    (void)hipEventRecord(r.e0, s);
    recs().push_back(r);
    open_rec = true;
}

void y2_prof_end(hipStream_t s) {
    if (!open_rec) return;
    (void)hipEventRecord(recs().back().e1, s);
    open_rec = false;
}

extern "C" int y2_prof_enable(int on) {
    if (on) { recs().clear(); pool_used = 0; }
    y2_prof_on = on ? 1 : 0;
    return Y2_OK;
}

extern "C" int y2_prof_count(void) { return (int)recs().size(); }

extern "C" int y2_prof_set_tag(int tag) { cur_tag = tag; return Y2_OK; }

extern "C" int y2_prof_get_tag(int i) { return (i < 0 || (size_t)i >= recs().size()) ? Y2_EINVAL : recs()[(size_t)i].tag; }

extern "C" int y2_prof_get(int i, char* name, int name_cap, float* ms, double* flops) {
    if (i < 0 || (size_t)i >= recs().size()) return Y2_EINVAL;
    const Rec& r = recs()[(size_t)i];
    if (name != nullptr && name_cap > 0) { strncpy(name, r.name, (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (flops != nullptr) *flops = r.flops;
    if (ms != nullptr) {
        hipError_t e = hipEventSynchronize(r.e1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, r.e0, r.e1);
        if (e != hipSuccess) return -(1000 + (int)e);
    }
    return Y2_OK;
}
