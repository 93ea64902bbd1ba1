// Library-level plumbing: version, error text, launch check.
#include "common.hpp"

#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace ultr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return ULTR_EHIP;
  }
  return ULTR_OK;
}

// ---- per-kernel timing -----------------------------------------------------------------------
bool g_profile_on = false;
struct ProfRec {
  const char* name;
  hipEvent_t e0, e1;
};
static std::vector<ProfRec*> g_recs;
static std::mutex g_prof_mu;

// stable storage for profile names composed at run time (template arguments, layer shape)
const char* profile_intern(const char* fmt, ...) {
  static std::map<std::string, std::string*> names;
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  auto it = names.find(buf);
  if (it == names.end()) it = names.emplace(buf, new std::string(buf)).first;
  return it->second->c_str();
}

void profile_begin(const char* name, hipStream_t s, void** handle) {
  ProfRec* r = new ProfRec();
  r->name = name;
  if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) {
    delete r;
    return;
  }
  (void)hipEventRecord(r->e0, s);
  *handle = r;
}
void profile_end(void* handle, hipStream_t s) {
  ProfRec* r = static_cast<ProfRec*>(handle);
  (void)hipEventRecord(r->e1, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_recs.push_back(r);
}

}  // namespace ultr

// Turn per-kernel timing on/off (clears what was collected).  Not for use under graph capture.
extern "C" int ultr_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(ultr::g_prof_mu);
  for (ultr::ProfRec* r : ultr::g_recs) {
    (void)hipEventDestroy(r->e0);
    (void)hipEventDestroy(r->e1);
    delete r;
  }
  ultr::g_recs.clear();
  ultr::g_profile_on = on != 0;
  return ULTR_OK;
}

// Synchronise and write "name count total_ms\n" lines (one per kernel) into buf; returns bytes written.
extern "C" int ultr_profile_report(char* buf, int buflen) {
  std::lock_guard<std::mutex> lk(ultr::g_prof_mu);
  std::map<std::string, std::pair<long long, double>> acc;
  for (ultr::ProfRec* r : ultr::g_recs) {
    if (hipEventSynchronize(r->e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r->e0, r->e1) != hipSuccess) continue;
    auto& a = acc[r->name];
    a.first += 1;
    a.second += ms;
  }
  int n = 0;
  for (auto& kv : acc) {
    const int w = snprintf(buf + n, n < buflen ? buflen - n : 0, "%s %lld %.6f\n", kv.first.c_str(), kv.second.first,
                           kv.second.second);
    if (w < 0 || n + w >= buflen) break;
    n += w;
  }
  return n;
}

extern "C" int ultr_host_device_pointer(void* host, void** device) {
  ULTR_REQUIRE(host && device, ULTR_EINVAL, "ultr_host_device_pointer: null pointer");
  *device = nullptr;
  if (hipHostGetDevicePointer(device, host, 0) != hipSuccess || !*device) {
    (void)hipGetLastError();
    ultr::set_error("ultr_host_device_pointer: %p is not mapped pinned host memory", host);
    *device = nullptr;
    return ULTR_EINVAL;
  }
  return ULTR_OK;
}

extern "C" int ultr_version(void) { return ULTR_VERSION; }
extern "C" const char* ultr_last_error(void) { return ultr::g_err; }
