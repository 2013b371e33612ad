// ugs_alloc.cpp - every device allocation of the library goes through ugs_malloc / ugs_free (ugs_dev.h).
//
// Default: hipMalloc / hipFree.
//
// UGS_GUARD_ALLOC=1 (debug, read once per process): every buffer is a mapping of its own made with the virtual-memory API -
// hipMemAddressReserve of [guard | pages | guard], hipMemCreate + hipMemMap of the pages only - and the pointer handed out is
// RIGHT-ALIGNED against the unmapped guard behind it (to UGS_GUARD_ALIGN bytes, default 16 = the widest per-lane load the kernels
// issue).  A kernel that reads or writes one element past what the host asked for then raises a GPU memory fault every time,
// instead of only when the allocator happens to have left the next page unmapped (the way to turn an
// out-of-bounds prefetch that aborts one run in six into a deterministic one).  UGS_GUARD_ALLOC=2 keeps the address range reserved
// after the free as well (use after free faults too; address space is never reused).
//
// UGS_ABORT_BT=<file>: a SIGABRT handler (installed when the library is loaded) that writes the backtrace of the ABORTING thread to
// <file> ("1" / "stderr": fd 2) and then chains to the handler that was there before (pytest's faulthandler, the default action).
// The ROCr runtime reports a GPU memory fault or a queue error by printing one line to fd 2 and calling abort() on one of ITS
// threads; glibc does the same for a corrupted heap; a test harness that captures fd 2 swallows the line, and a Python-level
// fault handler only shows the main thread.  This handler names the thread that died.
#include "ugs_host.h"
#include <execinfo.h>
#include <signal.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
struct GuardRec { void *va; size_t va_bytes; void *map_at; size_t map_bytes; hipMemGenericAllocationHandle_t h; int dev; size_t asked; const char *file; int line; void *user; };
std::mutex g_mu;
std::unordered_map<void *, GuardRec> g_live;
GuardRec g_freed[64];                 // the last released buffers (a fault on one of their pages is a use after free)
unsigned g_freed_n = 0;
int g_mode = -1;            // -1 unread, 0 off, 1 guard, 2 guard + keep the address range
size_t g_align = 16;
unsigned long long g_n_alloc = 0, g_n_free = 0, g_bytes_live = 0, g_bytes_peak = 0;

int guard_mode()
{
  if (g_mode < 0) {
    const char *e = getenv("UGS_GUARD_ALLOC");
    int m = (e && *e) ? atoi(e) : 0;
    if (m < 0 || m > 2) m = 0;
    const char *a = getenv("UGS_GUARD_ALIGN");
    if (a && *a) { long v = atol(a); if (v >= 1 && v <= 4096 && (v & (v - 1)) == 0) g_align = (size_t)v; }
    g_mode = m;
  }
  return g_mode;
}

hipError_t guard_malloc(void **out, size_t bytes, const char *file, int line)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess) return e;
  if (gran == 0) gran = 4096;
  const size_t want = bytes ? bytes : 1;
  const size_t user = (want + g_align - 1) / g_align * g_align;
  const size_t map_bytes = (user + gran - 1) / gran * gran;
  GuardRec r;
  r.dev = dev; r.asked = bytes; r.file = file; r.line = line; r.map_bytes = map_bytes; r.va_bytes = map_bytes + 2 * gran;
  e = hipMemAddressReserve(&r.va, r.va_bytes, gran, nullptr, 0);
  if (e != hipSuccess) return e;
  e = hipMemCreate(&r.h, map_bytes, &prop, 0);
  if (e != hipSuccess) { (void)hipMemAddressFree(r.va, r.va_bytes); return e; }
  r.map_at = (char *)r.va + gran;
  e = hipMemMap(r.map_at, map_bytes, 0, r.h, 0);
  if (e != hipSuccess) { (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.va, r.va_bytes); return e; }
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(r.map_at, map_bytes, &acc, 1);
  if (e != hipSuccess) { (void)hipMemUnmap(r.map_at, map_bytes); (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.va, r.va_bytes); return e; }
  void *p = (char *)r.map_at + (map_bytes - user);
  r.user = p;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[p] = r;
    ++g_n_alloc; g_bytes_live += map_bytes; if (g_bytes_live > g_bytes_peak) g_bytes_peak = g_bytes_live;
  }
  *out = p;
  return hipSuccess;
}

hipError_t guard_free(void *p)
{
  GuardRec r;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return hipErrorInvalidValue;
    r = it->second;
    g_live.erase(it);
    g_freed[g_freed_n++ % 64] = r;
    ++g_n_free; g_bytes_live -= r.map_bytes;
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != r.dev) (void)hipSetDevice(r.dev);
  (void)hipDeviceSynchronize();               // what hipFree does implicitly
  hipError_t e = hipMemUnmap(r.map_at, r.map_bytes);
  hipError_t e2 = hipMemRelease(r.h);
  hipError_t e3 = g_mode == 2 ? hipSuccess : hipMemAddressFree(r.va, r.va_bytes);
  if (cur != r.dev) (void)hipSetDevice(cur);
  return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : e3);
}
}  // namespace

hipError_t ugs_malloc_at(void **p, size_t bytes, const char *file, int line)
{
  if (guard_mode() == 0) return hipMalloc(p, bytes);
  return guard_malloc(p, bytes, file, line);
}

hipError_t ugs_free(void *p)
{
  if (!p) return hipSuccess;
  if (guard_mode() == 0) return hipFree(p);
  return guard_free(p);
}

// diagnostics for the tests: mode, allocations made / released, bytes mapped now / at the peak
extern "C" int ugs_debug_alloc_stats(unsigned long long out[5])
{
  std::lock_guard<std::mutex> lk(g_mu);
  out[0] = (unsigned long long)guard_mode(); out[1] = g_n_alloc; out[2] = g_n_free; out[3] = g_bytes_live; out[4] = g_bytes_peak;
  return UGS_OK;
}

// ---------------------------------------------------------------- ugs_h2d (ugs_host.h)
namespace {
constexpr size_t STAGE_BYTES = 8ull << 20;      // per slot
struct Stage {
  void *p[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false}; bool ready = false; unsigned next = 0;
  ~Stage() { /* released with the process: the device may be gone when thread-locals are destroyed */ }
};
constexpr int STAGE_DEVS = 16;
thread_local Stage g_stage[STAGE_DEVS];      // one pair of slots per (host thread, device): a thread that walks over the devices keeps them all
}  // namespace

hipError_t ugs_h2d(void *dst, const void *src, size_t bytes, hipStream_t st)
{
  if (!bytes) return hipSuccess;
  {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, src) == hipSuccess && (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged))
      return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);          // page-locked already: true DMA from the caller's buffer
    (void)hipGetLastError();                                                       // (plain pageable memory is "invalid value" to the query)
  }
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= STAGE_DEVS) {                                              // (no slots for such a device: the plain synchronous copy)
    if ((e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    return hipStreamSynchronize(st);
  }
  Stage &sg = g_stage[dev];
  if (!sg.ready) {                                                                 // first use of this device on this thread
    for (int k = 0; k < 2; ++k) {
      if (!sg.p[k] && (e = hipHostMalloc(&sg.p[k], STAGE_BYTES, hipHostMallocDefault)) != hipSuccess) return e;
      if (!sg.ev[k] && (e = hipEventCreateWithFlags(&sg.ev[k], hipEventDisableTiming)) != hipSuccess) return e;
      sg.busy[k] = false;
    }
    sg.ready = true; sg.next = 0;
  }
  const char *s = (const char *)src; char *d = (char *)dst;
  for (size_t off = 0; off < bytes; off += STAGE_BYTES) {
    const unsigned k = sg.next++ & 1u;
    const size_t n = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
    if (sg.busy[k] && (e = hipEventSynchronize(sg.ev[k])) != hipSuccess) return e; // the slot's previous chunk has left it
    memcpy(sg.p[k], s + off, n);
    if ((e = hipMemcpyAsync(d + off, sg.p[k], n, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    if ((e = hipEventRecord(sg.ev[k], st)) != hipSuccess) return e;
    sg.busy[k] = true;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------- UGS_KERNEL_LOG
int ugs_kernel_log = 0;
void ugs_before_launch(const char *kernel) { fprintf(stderr, "[ugs] kernel %.160s launched\n", kernel); fflush(stderr); }
void ugs_after_launch(const char *kernel, hipStream_t st)
{
  const hipError_t e = hipStreamSynchronize(st);
  fprintf(stderr, "[ugs] kernel %.160s %s\n", kernel, e == hipSuccess ? "done" : hipGetErrorString(e));
  fflush(stderr);
}

// ---------------------------------------------------------------- UGS_ABORT_BT
namespace {
struct sigaction g_old_abrt;
int g_bt_fd = -1;
volatile sig_atomic_t g_in_bt = 0;

void wr(int fd, const char *s) { ssize_t r = write(fd, s, strlen(s)); (void)r; }

void on_abort(int sig, siginfo_t *si, void *uc)
{
  if (!g_in_bt) {
    g_in_bt = 1;
    const int fd = g_bt_fd >= 0 ? g_bt_fd : 2;
    char line[256];
    snprintf(line, sizeof(line), "\n[ugs] SIGABRT on thread %ld of pid %d - backtrace of the aborting thread:\n", (long)gettid(), (int)getpid());
    wr(fd, line);
    void *frames[96];
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, fd);
    if (g_mode > 0) {
      // the guard allocator's table: the runtime's message above names the faulting PAGE - the buffer whose [user, end) lies right in front
      // of it was overrun, one that begins right behind it was underrun, a released one was used after its free
      wr(fd, "[ugs] guard allocator, live buffers (user pointer, end = first unmapped byte, bytes asked for, allocated at):\n");
      for (const auto &kv : g_live) {                      // (no lock: the process is dying; a torn read costs a garbled line)
        const GuardRec &r = kv.second;
        snprintf(line, sizeof(line), "  %p .. %p  %zu  %s:%d\n", r.user, (void *)((char *)r.map_at + r.map_bytes), r.asked, r.file ? r.file : "?", r.line);
        wr(fd, line);
      }
      wr(fd, "[ugs] guard allocator, the last released buffers:\n");
      for (unsigned k = 0; k < 64 && k < g_freed_n; ++k) {
        const GuardRec &r = g_freed[(g_freed_n - 1 - k) % 64];
        snprintf(line, sizeof(line), "  %p .. %p  %zu  %s:%d (released)\n", r.user, (void *)((char *)r.map_at + r.map_bytes), r.asked, r.file ? r.file : "?", r.line);
        wr(fd, line);
      }
    }
    wr(fd, "[ugs] /proc/self/maps lines of libamdhip64 / libhsa-runtime64 / librccl / libugs:\n");
    const int mfd = open("/proc/self/maps", O_RDONLY);
    if (mfd >= 0) {                                        // (no stdio, no malloc: the heap may be what is broken)
      static char buf[1 << 16];
      static char ln[1024];
      size_t ll = 0;
      ssize_t got;
      while ((got = read(mfd, buf, sizeof(buf))) > 0) {
        for (ssize_t i = 0; i < got; ++i) {
          const char c = buf[i];
          if (ll + 1 < sizeof(ln)) ln[ll++] = c;
          if (c == '\n') {
            ln[ll] = 0;
            if (strstr(ln, "r-xp") && (strstr(ln, "libamdhip64") || strstr(ln, "libhsa-runtime64") || strstr(ln, "librccl") || strstr(ln, "libugs"))) wr(fd, ln);
            ll = 0;
          }
        }
      }
      close(mfd);
    }
    fsync(fd);
  }
  // chain: the handler that was installed before this one (pytest's faulthandler), else the default action
  if (g_old_abrt.sa_flags & SA_SIGINFO) {
    if (g_old_abrt.sa_sigaction) { g_old_abrt.sa_sigaction(sig, si, uc); }
  } else if (g_old_abrt.sa_handler != SIG_DFL && g_old_abrt.sa_handler != SIG_IGN) {
    g_old_abrt.sa_handler(sig);
  }
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
}

__attribute__((constructor)) void ugs_install_abort_bt()
{
  { const char *k = getenv("UGS_KERNEL_LOG"); ugs_kernel_log = (k && *k && *k != '0') ? 1 : 0; }
  const char *e = getenv("UGS_ABORT_BT");
  if (!e || !*e) return;
  if (strcmp(e, "1") != 0 && strcmp(e, "stderr") != 0) g_bt_fd = open(e, O_WRONLY | O_CREAT | O_APPEND, 0644);
  void *warm[4];
  (void)backtrace(warm, 4);                               // (loads libgcc now, not inside the handler)
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_abort;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGABRT, &sa, &g_old_abrt);
}
}  // namespace
