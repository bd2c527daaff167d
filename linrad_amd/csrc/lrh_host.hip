// lrh_host.hip -- C ABI (include/linrad_hip.h) over the HIP kernels: context, device rings, tables, and the
// pointer bookkeeping the reference keeps in globals (citations per function).  Host code only; no CPU fallback:
// every stage launches HIP kernels and fails with LRH_EDEVICE when there is no usable GPU.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>
#include <vector>
#include <chrono>
#include <functional>
#include <map>
#include <atomic>
#include <mutex>

#include <exception>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/syscall.h>
#include <sys/stat.h>

#include "../../include/linrad_hip.h"
#include "lrh_kernels.hip.h"
#include "lrh_phase.h"

namespace lrh {
hipError_t launch_fft1(int log2n, const Fft1Args &a, int batch, hipStream_t st);
hipError_t launch_timf2(int log2n, const Timf2Args &a, int batch, hipStream_t st, const SumsqArgs *ss = nullptr, float *ss_part = nullptr, int *ss_run = nullptr);
hipError_t launch_sumsq_join(const SumsqArgs &a, const float *part, int run, hipStream_t st);
int timf2_grid(int log2n, int batch);
hipError_t launch_fft2(int log2n, const Fft2Args &a, int batch, hipStream_t st);
hipError_t launch_mix1_back(int log2n, const Mix1Args &a, int batch, hipStream_t st);
hipError_t launch_fft2_big(int log2n, const Fft2BigArgs &a, int batch, hipStream_t st, int steps = 3);
hipError_t launch_mix1_out(const Mix1OutArgs &a, int batch, hipStream_t st);
hipError_t launch_sumsq(const SumsqArgs &a, hipStream_t st);
hipError_t launch_slowsum(const SlowsumArgs &a, hipStream_t st);
hipError_t launch_powersum2(const Powersum2Args &a, hipStream_t st);
hipError_t launch_power_of(const float2 *src, float *dst, size_t n, hipStream_t st);
hipError_t launch_foldcorr(const FoldcorrArgs &a, int batch, hipStream_t st);
hipError_t launch_expand18(const unsigned char *packed, int ngroups, void *ring, int first_group, int group_mask, hipStream_t st);
hipError_t launch_waterfall(const WaterfallArgs &a, int nlines, hipStream_t st);
hipError_t launch_xypower(const XyArgs &a, hipStream_t st);
hipError_t launch_pol(const PolArgs &a, hipStream_t st);
hipError_t launch_realsplit(const RealSplitArgs &a, int batch, hipStream_t st);
hipError_t launch_timf2_net(const float2 *w, const float2 *s, int mask, int first, int count, float gain, float strong, float2 *dst, hipStream_t st);
hipError_t launch_blanker(const BlankArgs &a, int ring_words, hipStream_t st);
hipError_t launch_sellim2(const SellimArgs &a, hipStream_t st);
hipError_t launch_timf2_big(int log2n, const Timf2BigArgs &a, int batch, hipStream_t st);
hipError_t launch_span_copy(float *x, float *ring, int pbeg, int count, int mask, int to_ring, hipStream_t st);
hipError_t launch_span_copy2(float2 *x, float2 *ring, int first, int count, int mask, int to_ring, hipStream_t st);
hipError_t launch_blockpower(const BlockpowerArgs &a, int nblocks, hipStream_t st);
hipError_t launch_fft3(int log2n, const Fft3Args &a, int batch, hipStream_t st);
hipError_t launch_spur(const SpurArgs &a, hipStream_t st);
hipError_t launch_corrsum(const CorrArgs &a, hipStream_t st);
hipError_t launch_spur_patch(const SpurPatchArgs &a, int nspurs, int ngroups, hipStream_t st);
hipError_t launch_spur_acquire(const SpurArgs &a, int pnt, int *result, hipStream_t st);
hipError_t launch_mix2_back(int log2n, const Mix2Args &a, int batch, hipStream_t st);
}  // namespace lrh
using namespace lrh;

#define PI_L 3.1415926535897932
#define NATLOG 2.718281828459045
#define FFT2_WATERFALL_ZERO 0.012
#define LRH_NSTAGE 4
#define LRH_BLN_PARTIALS 256
#define LRH_NOUT 32                 /* read-back slots (export_impl) */
#define LRH_OUT_SLOT_BYTES (1u << 19)
#define LRH_MAX_HANDLES 7          /* handle 0 = the caller's own thread; 1..6 = THREAD_FFT1B1..6 (MAX_FFT1_THREADS, thrdef.h:107; gpu_handle_number, wcw.c:500) */

// LRH_CRASH_TRACE=1 (the test suite sets it): a fatal signal prints the C stack of the thread it was raised on, AFTER whatever handler was
// installed before (Python's faulthandler under pytest), so the native frames are the last thing in the log.  Diagnostics only: without the
// switch the library installs nothing.  (Round 5's GPU suite died of a SIGABRT raised on a thread of the HIP runtime with no message at all.)
namespace {
// LRH_ALLOC_LOG=2: allocations / releases / context opens and closes go to a ring in memory instead of stderr (no change of timing); the crash
// handler prints it, so a faulting address can be matched with the buffer -- live or released -- it lies in
struct AllocRec { char op; const void *p; size_t bytes; };
constexpr int ALLOC_RING = 1 << 15;
AllocRec g_alloc_ring[ALLOC_RING]; std::atomic<unsigned> g_alloc_n{0};
int alloc_log_mode() { static const int m = getenv("LRH_ALLOC_LOG") ? atoi(getenv("LRH_ALLOC_LOG")) : 0; return m; }
void alloc_note(char op, const void *p, size_t bytes)
{
  const int m = alloc_log_mode();
  if (m == 1) fprintf(stderr, "LRH_%s %p %zu\n", op == 'D' ? "ALLOC dev" : op == 'H' ? "ALLOC host" : op == 'd' ? "FREE dev" : op == 'h' ? "FREE host" : op == 'O' ? "OPEN ctx" : "CLOSE ctx", p, bytes);
  else if (m == 2) { const unsigned i = g_alloc_n.fetch_add(1); g_alloc_ring[i % ALLOC_RING] = {op, p, bytes}; }
}
struct sigaction g_crash_prev[NSIG];
int g_crash_fd[3] = {2, -1, -1};          // stderr, LRH_CRASH_TRACE_FD (pytest captures fd 2: its faulthandler writes to a duplicate of the real one, so do we), LRH_CRASH_TRACE_FILE
void crash_write(const char *s) { for (int fd : g_crash_fd) if (fd >= 0 && write(fd, s, strlen(s)) < 0) {} }
void crash_handler(int sig, siginfo_t *si, void *uc)
{
  const struct sigaction prev = g_crash_prev[sig];
  if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction) prev.sa_sigaction(sig, si, uc);      // faulthandler: dumps, puts SIG_DFL back, raises (held back while we are in here)
  else if (!(prev.sa_flags & SA_SIGINFO) && prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN) prev.sa_handler(sig);
  char line[160];
  snprintf(line, sizeof line, "\n=== liblinrad_hip crash trace: signal %d, thread %ld (process %d) ===\n", sig, (long)syscall(SYS_gettid), (int)getpid());
  crash_write(line);
  { // what was last written to fd 2 when that is a regular file: pytest's capture file swallows the runtime's own last words
    // ("Memory access fault by GPU node-..." of the ROCr fault handler) -- read them back through /proc
    struct stat st_;
    if (fstat(2, &st_) == 0 && S_ISREG(st_.st_mode) && st_.st_size > 0) {
      const int f2 = open("/proc/self/fd/2", O_RDONLY);
      if (f2 >= 0) {
        static char tailbuf[4097];
        const off_t from = st_.st_size > 4096 ? st_.st_size - 4096 : 0;
        const ssize_t got = pread(f2, tailbuf, 4096, from);
        if (got > 0) { tailbuf[got] = 0; for (int k = 1; k < 3; k++) if (g_crash_fd[k] >= 0) { if (write(g_crash_fd[k], "--- tail of the captured stderr ---\n", 36) < 0 || write(g_crash_fd[k], tailbuf, (size_t)got) < 0 || write(g_crash_fd[k], "\n--- end of captured stderr ---\n", 33) < 0) {} } }
        close(f2);
      }
    }
  }
  void *bt[48]; const int n = backtrace(bt, 48);
  for (int fd_ : g_crash_fd) if (fd_ >= 0) backtrace_symbols_fd(bt, n, fd_);
  const int fd = open("/proc/self/maps", O_RDONLY);              // where the runtime libraries sit (two HIP runtimes in one process is a finding of its own)
  if (fd >= 0) {
    static char buf[1 << 16]; size_t have = 0; ssize_t r;
    while ((r = read(fd, buf + have, sizeof buf - 1 - have)) > 0) {
      have += (size_t)r; buf[have] = 0;
      char *ls = buf, *nl;
      while ((nl = strchr(ls, '\n'))) {
        *nl = 0;
        if (strstr(ls, " r-xp ") && (strstr(ls, "libamdhip64") || strstr(ls, "libhsa-runtime64") || strstr(ls, "liblinrad_hip"))) { crash_write(ls); crash_write("\n"); }
        ls = nl + 1;
      }
      have = strlen(ls); memmove(buf, ls, have + 1);
    }
    close(fd);
  }
  if (alloc_log_mode() == 2) {
    const unsigned n = g_alloc_n.load(), first = n > ALLOC_RING ? n - ALLOC_RING : 0;
    crash_write("allocation journal (D/H device/host allocation, d/h release, O/C context open/close):\n");
    for (unsigned i = first; i < n; i++) { const AllocRec &r = g_alloc_ring[i % ALLOC_RING]; snprintf(line, sizeof line, "%c %p %zu\n", r.op, r.p, r.bytes); crash_write(line); }
  }
  crash_write("=== end of crash trace ===\n");
  signal(sig, SIG_DFL); raise(sig);
}
struct CrashTraceInstall {
  CrashTraceInstall() {
    const char *e = getenv("LRH_CRASH_TRACE");
    if (!e || !atoi(e)) return;
    if (const char *f = getenv("LRH_CRASH_TRACE_FD")) { const int fd = atoi(f); if (fd > 2 && fcntl(fd, F_GETFD) != -1) g_crash_fd[1] = fd; }
    if (const char *f = getenv("LRH_CRASH_TRACE_FILE")) g_crash_fd[2] = open(f, O_WRONLY | O_CREAT | O_APPEND, 0644);
    void *warm[4]; backtrace(warm, 4);                             // loads libgcc's unwinder now, not inside the handler
    for (int sig : { SIGABRT, SIGSEGV, SIGBUS, SIGFPE, SIGILL }) {
      struct sigaction sa; memset(&sa, 0, sizeof sa);
      sa.sa_sigaction = crash_handler; sa.sa_flags = SA_SIGINFO | SA_ONSTACK; sigemptyset(&sa.sa_mask);
      sigaction(sig, &sa, &g_crash_prev[sig]);
    }
  }
} g_crash_trace_install;
}  // namespace

// Device memory.  LRH_GUARD=1 (diagnostics; tests/test_gpu_stress.py): every allocation gets an address range of its own through the
// virtual-memory calls, with an unmapped granule on either side and the buffer pushed against the upper one (16-byte steps), so that an
// access past either end of a buffer is a page fault at the faulting kernel instead of a silent read of whatever the neighbour holds --
// the pool's GPUs have no address sanitizer.  LRH_GUARD=2: the same without the 256 spare bytes dev_alloc adds.
namespace {
struct GuardRec { void *base; size_t va, mapped; hipMemGenericAllocationHandle_t h; };
std::mutex g_guard_mtx; std::map<void *, GuardRec> g_guard;
int guard_mode() { static const int m = getenv("LRH_GUARD") ? atoi(getenv("LRH_GUARD")) : 0; return m; }
}
// LRH_POISON=1 (diagnostics; tests/test_gpu_stress.py): fresh device memory is filled with 0x7f bytes (floats 3.4e38, ints 2.1e9) before the
// caller sees it -- a kernel that reads what nobody wrote then says so with a NaN / a wild index at once, instead of depending on what the
// previous owner of the pages left behind (a fresh process mostly gets zeroes, the 150th context of a test session does not)
static bool poison_on() { static const bool on = getenv("LRH_POISON") && atoi(getenv("LRH_POISON")); return on; }
static hipError_t lrh_dev_malloc_raw(void **p, size_t bytes);
static hipError_t lrh_dev_malloc(void **p, size_t bytes)
{
  const hipError_t e_ = lrh_dev_malloc_raw(p, bytes);
  if (e_ == hipSuccess && poison_on() && bytes) { hipMemset(*p, 0x7f, bytes); hipDeviceSynchronize(); }
  return e_;
}
static bool alloc_log() { return alloc_log_mode() != 0; }   // diagnostics: every device allocation / release on stderr
static hipError_t lrh_dev_malloc_raw(void **p, size_t bytes)
{
  if (!guard_mode()) { const hipError_t e_ = hipMalloc(p, bytes); if (alloc_log()) alloc_note('D', e_ == hipSuccess ? *p : nullptr, bytes); return e_; }
  int dev = 0; hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return e;
  hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gran = 0; if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
  if (bytes == 0) bytes = 16;
  GuardRec r; r.mapped = (bytes + gran - 1) / gran * gran; r.va = r.mapped + 2 * gran;
  if ((e = hipMemAddressReserve(&r.base, r.va, gran, nullptr, 0)) != hipSuccess) return e;
  if ((e = hipMemCreate(&r.h, r.mapped, &prop, 0)) != hipSuccess) { hipMemAddressFree(r.base, r.va); return e; }
  char *lo = (char *)r.base + gran;
  if ((e = hipMemMap(lo, r.mapped, 0, r.h, 0)) != hipSuccess) { hipMemRelease(r.h); hipMemAddressFree(r.base, r.va); return e; }
  hipMemAccessDesc d; memset(&d, 0, sizeof d); d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemSetAccess(lo, r.mapped, &d, 1)) != hipSuccess) { hipMemUnmap(lo, r.mapped); hipMemRelease(r.h); hipMemAddressFree(r.base, r.va); return e; }
  char *user = lo + r.mapped - (bytes + 15) / 16 * 16;
  { std::lock_guard<std::mutex> lk(g_guard_mtx); g_guard[user] = r; }
  *p = user;
  return hipSuccess;
}
static hipError_t lrh_host_malloc(void **p, size_t bytes)
{
  const hipError_t e_ = hipHostMalloc(p, bytes);
  if (e_ == hipSuccess && poison_on()) memset(*p, 0x7f, bytes);
  if (alloc_log()) alloc_note('H', e_ == hipSuccess ? *p : nullptr, bytes);
  return e_;
}
template <typename T> static hipError_t lrh_host_malloc(T **p, size_t bytes) { return lrh_host_malloc((void **)p, bytes); }
static hipError_t lrh_host_free(void *p) { if (alloc_log()) alloc_note('h', p, 0); return hipHostFree(p); }
static hipError_t lrh_dev_free(void *p)
{
  if (!p) return hipSuccess;
  GuardRec r; bool found = false;
  { std::lock_guard<std::mutex> lk(g_guard_mtx); auto it = g_guard.find(p); if (it != g_guard.end()) { r = it->second; g_guard.erase(it); found = true; } }
  if (alloc_log()) alloc_note('d', p, 0);
  if (!found) return hipFree(p);
  hipDeviceSynchronize();
  size_t gran = (r.va - r.mapped) / 2;
  hipMemUnmap((char *)r.base + gran, r.mapped); hipMemRelease(r.h);
  return hipMemAddressFree(r.base, r.va);
}

struct ProfEntry { double ms = 0; long n = 0; };
struct ProfPending { std::string name; hipEvent_t e0, e1; };

struct lrh_ctx {
  // Linrad calls the stage functions from its stage threads (wideband_dsp and up to six fft1_b workers, timf2_routine,
  // second_fft, narrowband_dsp; thrdef.h:49-107): every entry point takes this lock while it does its bookkeeping and
  // enqueues its device work.  The work itself is asynchronous, so the lock is held for microseconds; device-side order
  // is the order of the calls (one main stream), which the caller's events already make the reference's order.
  std::recursive_mutex mtx;
  // fft1_b workers: handle h >= 1 launches on its own stream so that transforms of different workers overlap
  hipStream_t hstream[LRH_MAX_HANDLES] = {}; hipEvent_t hev[LRH_MAX_HANDLES] = {}, hev_start = nullptr; std::atomic<bool> hpending[LRH_MAX_HANDLES] = {}; std::atomic<bool> hread[LRH_MAX_HANDLES] = {};
  // what the workers' transforms have read of timf1 since the producer last waited for them: [rd_lo, rd_lo + rd_span) in ring bytes (mtx_w).
  // lrh_timf1_write_async waits for the workers' events only when its copy would touch that span -- once per lap of the ring, not once per block
  bool rd_any = false; long long rd_lo = 0, rd_span = 0;
  std::mutex mtx_w; bool worker_fast = true;   // the workers' calls take this small lock (wparked, rd_*) instead of the context's; LRH_WORKER_FAST=0: they launch themselves, on their own streams
  lrh_config cfg;
  bool opening = false;       // inside lrh_open: dev_alloc's memsets are waited for once, before the tables go up
  int N1, I1, M1, N2, I2, M2, Nm, Im, Mm, mix1_n;
  int timf2_mode;
  int lowlevel_points = 0;   // liminfo[i]==0 count of the table in force (timf2.c:37-52)
  int mix_cap = 0;           // transforms per mix1 launch the scratch buffers hold
  float *d_blockpower = nullptr;
  // fft3 / mix2
  int N3 = 0, I3 = 0, M3 = 0, Nm2 = 0, Im2 = 0, Mm2 = 0;
  float *d_window3 = nullptr, *d_bgfilt = nullptr; float2 *d_tw3 = nullptr, *d_twm2 = nullptr, *d_fft3 = nullptr, *d_baseb = nullptr, *d_mix2_scratch = nullptr;
  std::vector<float> h_window3_ref;
  int xcd_mask = 2;          // bit 0 fft1, 1 timf2, 2 fft2: XCD-aware block order (tuning knob LRH_XCD_MASK)
  hipStream_t stream = nullptr;      // main stream: every API call is ordered on it
  hipStream_t stream_in = nullptr;   // producer copies of lrh_timf1_write_async
  hipEvent_t ev_in = nullptr, ev_fft1_read = nullptr; std::atomic<hipEvent_t> ev_fft1_read_cur{nullptr}; std::atomic<bool> read_alias_wanted{false}, producer_seen{false}; hipEvent_t ev_in_guard = nullptr; std::atomic<bool> in_pending{false}, fft1_read_valid{false};
  std::mutex mtx_in;                  // the producer side (lrh_timf1_write_async / _wait) has a lock of its own: an input thread is never held up by a stage call that sleeps on the staging ring
  // Producer copies out of a page-locked arena are noted here and issued merged -- by the producer once in_merge_bytes have gathered, by the next
  // reader of timf1 (wait_for_input) or by lrh_timf1_write_wait / lrh_sync whichever comes first: a receiver that hands over one fft1 block per
  // call (32 kB at fft1_size 16384) otherwise makes 7680 hipMemcpyAsync calls per 63 Msamples, and every one of them holds the runtime's locks
  // against the stage threads' own calls (profiles/r06_glue_trace.txt).  Under mtx_in.  LRH_IN_MERGE_KB (0: every call copies at once)
  struct InSpan { const char *src; int off, nbytes; };
  std::vector<InSpan> in_spans; size_t in_span_bytes = 0, in_merge_bytes = 256u << 10;
  // ... and a reader waits for the copies that carry ITS samples, not for everything the producer has queued by then (a producer that runs ahead
  // -- a file played at full speed -- otherwise puts every transform behind a train of copies it does not need): after every issue an event
  // goes into this ring with the count of bytes issued so far and the ring offset they end at
  struct InMark { hipEvent_t ev = nullptr; long long total = -1; };
  InMark in_marks[32]; unsigned in_mark_n = 0; long long in_total = 0; int in_off_end = 0;
  hipStream_t stream_sel = nullptr;  // the limiter kernels: one workgroup for ~0.5 ms, kept off the side stream (the blanker of the next round queues there)
  hipStream_t stream3 = nullptr;     // upload stream: mix1 phase tables of the lagged schedule travel a round ahead of their kernels
  hipStream_t stream2 = nullptr;     // side stream for the bandwidth-bound small kernels inside lrh_wideband_dsp
  hipStream_t cur = nullptr;         // stream the stage functions launch on (== stream outside the pipelined driver)
  hipEvent_t ev_fft1 = nullptr, ev_timf2 = nullptr, ev_timf2b = nullptr, ev_blank = nullptr, ev_blank2[2] = {nullptr, nullptr}, ev_fft2 = nullptr, ev_side = nullptr, ev_ps2 = nullptr, ev_sumsq[2] = {nullptr, nullptr};
  bool split_fft2_tail = false;      // inside the two-stream schedule: powersum2 / waterfall go to the side stream
  // fft1_c's power sums inside k_timf2 (lrh_wideband_dsp, sin^2 window): fft1_c parks its arguments here, make_timf2
  // picks them up, the join of split groups and the slow average follow from ss_queue
  bool fuse_sumsq = true;            // LRH_FUSE_SUMSQ=0: separate k_sumsq pass
  bool sums_on_main = false;         // LRH_SUMS_MAIN (default: on for fft2_size <= 16384), see lrh_wideband_dsp
  int spare_cus = 0;                 // LRH_SPARE_CUS: see persistent_grid (used while lrh_wideband_limiter is installed; measured: no gain, 8 spare units cost 7 % of k_fft1 / k_timf2)
  bool ss_defer = false, ss_have = false; SumsqArgs ss_args; int ss_run = 1;
  // the same for fft1_b itself: inside lrh_wideband_dsp (fft1_size 16384, sin^2 window, int16 I/Q) its launch is parked and make_timf2
  // runs forward transform, sums and weak stream as one kernel (k_fft1w); any other reader of fft1_float issues the parked launch first
  bool f1_defer = false, f1_have = false, fuse_fft1 = true, fuse_fft1_forced = false; Fft1Args f1_args; int f1_batch = 0;   // fuse_fft1_forced: LRH_FUSE_FFT1=1 given (tests: the fused kernel whatever the batch)
  bool fuse_real = true;             // LRH_FUSE_REAL=0: real input through k_fft1<REAL> + k_realsplit + k_timf2 as before round 5
  bool fuse_v = true;                // LRH_FFT1V=0: k_fft1w (round 3) instead of k_fft1v where both exist (fft1_size 16384, int16)
  float2 *d_filtercorr_v = nullptr;  // the filter correction in k_fft1v's thread order (upload_filtercorr)
  bool f1_is_big = false; Fft1BigArgs f1_big;   // fft1_size 32768: the column step has run, the row step is what is parked (k_fft1r_t2c takes it)
  std::vector<int> fft2_keep_lo, fft2_keep_hi;   // per fft2 ring slot: the band lrh_make_fft2 stored (cfg.fft2_float_sparse)
  bool corr_on = false; int slowcorr_tot_avgnum = 0; float2 *d_xspec = nullptr, *d_corrsum = nullptr, *d_slowcorr = nullptr; double2 *d_slowcorr_tot = nullptr;   // lrh_set_correlation
  lrh_exchange_fn xfn = nullptr; void *xuser = nullptr;     // lrh_set_exchange: collectives of two coupled channels inside lrh_wideband_dsp
  const float2 *xy_own_src = nullptr;                       // set by dsp_coupled around lrh_fft2_xy_finish: the own channel's transforms where they lie in the fft2 ring
  bool timf2_primed = false;      // a transform has gone through make_timf2: the next one has an overlap partner
  // the fused kernels rebuild that partner from the timf1 ring (one block behind the first of the call): only right while the calls walk
  // the ring without a gap and the tables have not changed in between
  std::atomic<bool> f1_end_valid{false};      // f1_end is where the previous handle-0 lrh_fft1_b stopped, under the tables in force now
  int f1_end = 0;                 // frame index
  bool f1_cont = false;           // the parked call starts where the previous one stopped
  float *d_ss_part = nullptr; size_t ss_part_stride = 0; int ss_flip = 0;   // two halves, alternating per fused launch: the join of
                                                                             // round k (side stream) may still read while timf2(k+1) writes
  std::vector<std::function<int(lrh_ctx *)>> ss_queue;
  float ch2_c1 = 1.0f, ch2_c2 = 0.0f;   // lrh_set_ch2_phasing
  // two coupled RF channels (cfg.blanker_channels == 2): summed power ring, exchange buffers, state between the calls
  float *d_pwr_sum = nullptr, *d_xbuf = nullptr, *d_xstat = nullptr;
  float2 *d_xweak = nullptr, *d_tf_partner = nullptr; int x_span = 0, xw_count = 0;   // linear blanker on two coupled channels: LRH_X_WEAK, the partner's samples in ring places
  float2 *d_net = nullptr; size_t net_cap = 0;      // staging of lrh_export_timf2_net
  float2 *d_fft1net = nullptr; int fft1net_cap = 0; // staging of lrh_export_fft1_net: bare transforms, [pow2 >= batch][N1]
  float2 *d_xpol = nullptr; float2 pol_wa = {1.f, 0.f}, pol_wb = {0.f, 0.f}; bool pol_set = false; int pol_batch = 0;   // LRH_X_POL [2][max_fft3n][Nm2]; pg.c1..c3
  float2 *d_xbins = nullptr; float4 *d_xypower = nullptr, *d_xysum = nullptr, *d_xysum_alt = nullptr;   // LRH_X_BINS [2][max_fft2n][N2]; TWOCHAN_POWER rings
  int x_pbeg = 0, x_count = -1; bool fin_pending = false; BlankArgs fin_args;
  int dbg_stamp = 0, dbg_bln = 0;    // LRH_STAMP / LRH_BLN_DEBUG, read once in lrh_open
  // four-step fft2 in spans (LRH_FFT2_SPAN transforms each): the column step of span s+1 runs on the main stream beside the row step of
  // span s on a stream of its own, and the scratch of a span (67 MB at 128 transforms of 65536) is re-used every third span: what the
  // column step writes the row step reads while it is still in the 256 MB Infinity Cache, and a re-used line never reaches the HBM
  int fft2_span = 0; hipStream_t stream_f2 = nullptr; hipEvent_t ev_f2c[3] = {}, ev_f2r[3] = {};
  int env_fft2_run = 0, env_fft2_cols_run = 0;   // LRH_FFT2_RUN / LRH_FFT2_COLS_RUN: transforms per workgroup (0: automatic)
  unsigned long long *d_stamps = nullptr;
  bool early_upload = true;          // LRH_EARLY_UPLOAD=0: phase tables in stream order even when the kernels are parked
  // one-round-late schedule kept across calls: the parked launches (blanker / fft2 + mix1 of the last round) of the previous call
  std::vector<std::function<int(lrh_ctx *)>> pend_b, pend_t; bool pend = false, pend_tail_flushed = false; int pend_batch = 0, in_dsp = 0; double host_cpu_ms_wait = 0;
  bool persist = true;               // LRH_PERSIST=0: every call drains its pipeline before it returns
  bool pipeline_forced = false;      // LRH_PIPELINE given: no automatic choice by batch size
  int pipeline = 2;                  // LRH_PIPELINE: 0 serial, 1 two streams, 2 two streams with blanker / fft2 / mix1 one round behind
  // Deferred launches (schedule 2): while `rec` is set the stage functions do their host bookkeeping at once but append
  // their device work here; lrh_wideband_dsp replays it later, on the stream it chooses.
  std::vector<std::function<int(lrh_ctx *)>> *rec = nullptr;
  bool ph_pending[LRH_NSTAGE] = {};  // staging slot handed to a deferred upload that has not been replayed yet
  hipEvent_t ev_tail = nullptr;
  // Every event record and every wait is a packet the queue works off one after the other, ~4 us each with the next kernel held behind it
  // (rocprofv3 timeline, profiles/r04_timeline.txt): the two-stream schedule does not record a second event where one already marks the
  // same point of the main stream.  last_main_ev: an event recorded on the main stream with nothing enqueued there since (set and
  // consumed within a few lines of each other, never carried across calls); ev_tail_cur: the event that stands for ev_tail.
  hipEvent_t last_main_ev = nullptr, ev_tail_cur = nullptr;
  double ph_last_wait_us = 0;     // how long mix1_run last waited for a staging slot
  // Small rounds (serial order): blanker + fft2 + mix1 (+ fft3 / mix2) of a round are a chain of short kernels the next round's fft1 and
  // timf2 do not wait for; they go to the side stream and the main stream carries on.  st_n tails issued so far, ev_st[n & 1] behind tail n;
  // any entry point other than lrh_wideband_dsp first orders the main stream behind the newest tail (LRH_ENTER).  LRH_SIDE_TAIL=0: off.
  hipEvent_t ev_st[2] = {nullptr, nullptr}; unsigned st_n = 0; bool st_pending = false, st_on = true;
  // narrowband stream of the two-stream schedules: mix1 / fft3 / mix2 of a round -- a handful of small kernels, 66 us one after the other --
  // run here beside the next round's fft1 instead of holding the main stream (LRH_NARROW_STREAM=0: on the main stream as before)
  hipStream_t stream_nb = nullptr; hipEvent_t ev_f2done = nullptr, ev_nb = nullptr; bool nb_split = true, nb_pending = false; hipStream_t nb_keep = nullptr;
  // ev_nb: the newest of ev_nb_ring (an alias).  The next fft2 only has to wait for the narrowband group whose fft2 slots it is about to
  // overwrite: with a ring of several rounds that group ended long ago, and the newest one -- starved beside k_fft1v, it ends 20-50 us after
  // timf2s -- no longer holds fft2 up.  nb_first_total[j]: transforms fft2 had written when group j's first slot was written; f2_total: now.
  hipEvent_t ev_nb_ring[4] = {nullptr, nullptr, nullptr, nullptr}; bool nb_ev_valid[4] = {false, false, false, false};
  unsigned nb_seq = 0; long nb_first_total[4] = {0, 0, 0, 0}, f2_total = 0; bool nb_join_pending = false;
  unsigned char *d_pack18 = nullptr; size_t pack18_cap = 0;   // staging for lrh_timf1_write_packed18
  float2 *d_foldcorr = nullptr, *d_unitcorr = nullptr;   // I/Q mirror-image calibration (lrh_set_foldcorr); unit filter table for the bare transform
  bool fft2_fused = false;           // waterfall power sums formed inside k_fft2 (fft2_power ring then rebuilt on export)
  char err[256] = "";
  // device tables
  float *d_window1 = nullptr, *d_invwin1 = nullptr, *d_window2 = nullptr, *d_fqwin = nullptr, *d_yfac = nullptr;
  float *d_mixwin = nullptr, *d_sin2win = nullptr, *d_cos2win = nullptr; int Xm = 0;   // crossover-window mix1 (prepare_mixer, buf.c:55-111)
  std::vector<float> h_mixwin, h_sin2win, h_cos2win;
  float *d_bbfir = nullptr; int bbfir_pts = 0;              // bg.mixer_mode = 2 (lrh_set_basebraw_fir); nullptr: mixer_mode 1
  // the search for new spurs (lrh_spur_search_config): sums and search spectrum on the device, the cleanup on a stream of its own
  float *d_ss_sum = nullptr, *d_ss_spec_base = nullptr, *d_ss_min = nullptr, *d_ss_out = nullptr; int ss_first = 0, ss_last = 0, ss_counter = 0, ss_completed = 0;
  hipStream_t stream_ss = nullptr; hipEvent_t ev_ss_in = nullptr, ev_ss_done = nullptr; bool ss_busy = false;
  float *d_mix2win = nullptr, *d_sin2win2 = nullptr, *d_cos2win2 = nullptr; int Xm2 = 0;   // ... and mix2's (THIRD_FFT_SINPOW neither 0 nor 2, mix2.c:177-216)
  std::vector<float> h_mix2win, h_sin2win2, h_cos2win2;
  float2 *d_filtercorr = nullptr, *d_tw1 = nullptr, *d_tw2 = nullptr, *d_twm = nullptr, *d_tw2a = nullptr, *d_tw2b = nullptr, *d_fft2_scratch = nullptr;
  // fft1_size 32768: four-step fft1 / timf2 (tables of size 256 / 128; scratch per fft1_b handle and for timf2, grown on demand)
  float2 *d_tw1a = nullptr, *d_tw1b = nullptr, *d_fft1_scratch[8] = {}, *d_timf2_scratch = nullptr; size_t fft1_scratch_cap[8] = {}, timf2_scratch_cap = 0;
  bool fft1_big = false;
  unsigned int *d_pack_cur = nullptr, *d_pack_prev = nullptr;
  int *d_wf_itab = nullptr;
  // device rings
  short2 *d_timf1 = nullptr;
  float2 *d_fft1 = nullptr; float *d_sumsq = nullptr, *d_slowsum = nullptr;
  float2 *d_timf2w = nullptr, *d_timf2s = nullptr; float *d_pwr = nullptr; unsigned int *d_blnbits = nullptr;   // timf2 kept planar on the device
  float2 *d_fft2 = nullptr; float *d_power2 = nullptr, *d_powersum2 = nullptr, *d_powersum2_alt = nullptr, *d_wf_scratch = nullptr;
  int16_t *d_waterf = nullptr;
  float2 *d_timf3 = nullptr, *d_mix_scratch = nullptr;
  float *d_ph = nullptr;              // phase tables [LRH_NSTAGE][2][max_fft2 batch][half]
  BlankState *d_bst = nullptr; float *d_partials = nullptr; float4 *d_bln_tiles = nullptr; int *d_bln_counts = nullptr;
  unsigned long long *d_bln_wbusy = nullptr; int4 *d_bln_wstate = nullptr;     // tile-parallel walk of the calibrated blanker (k_blank_walk_*)
  // linear ("clever") blanker: tables of lrh_set_blanker_tables, per-sample flags and candidate bit words
  bool clever_on = false; lrh_blanker_tables bt{}; float *d_bt_refpulse = nullptr, *d_bt_phasefunc = nullptr; int *d_bt_pulindex = nullptr;
  unsigned char *d_bln_flag = nullptr; unsigned long long *d_bln_cand = nullptr;
  lrh_sellim wl_par{}; bool wl_on = false, wl_fft2 = false; int wl_cnt1 = 0, wl_cnt2 = 0; std::vector<float> wl_desired;   // lrh_wideband_limiter
  float *d_sel_reg = nullptr; size_t sel_reg_cap = 0;
  float *d_sel_ftmp = nullptr, *d_sel_desired = nullptr, *d_sel_bigb = nullptr, *d_sel_bigg = nullptr; float sel_desired_totsum = 0; std::vector<float> h_sel_desired;   // fftf_tmp of fft2_update_liminfo; calibration of the amplitude factor
  int *d_clv_start = nullptr, *d_clv_ext = nullptr, *d_clv_ctl = nullptr, *d_clv_bk_pos = nullptr, *d_clv_dbg = nullptr; unsigned long long *d_clv_logged = nullptr; float *d_clv_bk_pwr = nullptr; float2 *d_clv_bk_tf = nullptr; float *d_clv_bk_pwo = nullptr; float2 *d_clv_bk_ty = nullptr;
  // deferred schedule of lrh_wideband_dsp: the search of a round is issued a round late, its resume point comes back through a pinned slot
  // and the rest of that blanker call (statistics, dumb blanker) is issued when the next call -- which starts at the resume point -- comes
  int clv_split = 0; hipEvent_t clv_ev_t2 = nullptr;   // LRH_CLEVER_SPLIT=1: candidate bits and region list a round ahead of the replay (clv_ev_t2: this round's make_timf2, set by the deferred schedule); measured: slower
  int clv_first = 0;    // LRH_CLEVER_FIRST=1: the deferred search runs ahead of the round's forward transform instead of beside it (measured: slower)
  bool clv_wait = false, clv_issued = false, clv_defer = false; hipEvent_t ev_clv = nullptr, ev_amp = nullptr; int *h_clv_out = nullptr; float *d_clv_amp = nullptr; int clv_amp_seq = 0;
  struct { BlankArgs a; CleverArgs ca; int pbeg; float lowlevel; } clv_late;
  size_t clv_cap = 0; int clv_max_regions = 0; bool clever_force_serial = false;   // region list / backup of the span, grown on demand
  // host tables (reference layouts, for lrh_get_table)
  std::vector<float> h_window1_ref, h_invwin1_ref, h_window2, h_fqwin, h_filtercorr, h_desired, h_yfac;
  std::vector<unsigned int> h_pack;
  bool have_liminfo = false;
  // spurs being tracked (lrh_spur_config / lrh_spur_set): loop state and histories on the device, k_spur between fft2 and its power sums
  float2 *spur_ring = nullptr; int spur_nx = 0, spur_maxn = 0; float spur_ff = 0;   // the transforms the spurs are taken from: fft2, or fft1 with the second fft off
  int spur_max = 0, spur_n = 0, spur_speknum = 0; lrh_spur *d_spurs = nullptr; float *d_spur_table = nullptr, *d_spur_signal = nullptr, *d_spur_spectra = nullptr; int *d_spur_ind = nullptr, *d_spur_touched = nullptr;
  // selective limiter on the device (lrh_fft1_update_liminfo): the reference's liminfo / old_liminfo / liminfo_wait / fftt_tmp
  float *d_liminfo = nullptr, *d_old_liminfo = nullptr, *d_sel_tmp = nullptr; unsigned char *d_sel_wait = nullptr; SellimState *d_sel_st = nullptr;
  // weak-bin counts come back through a ring of pinned slots; without exact_stats the one installed is two updates old, so
  // the host never waits for a step it has only just enqueued
  int *h_sel_low = nullptr; hipEvent_t ev_sel = nullptr, ev_sel_slot[3] = {}; unsigned sel_seq = 0; bool sel_pending = false;
  // the limiter runs on the side stream, beside whatever the main stream still has queued: behind the last k_timf2 (which reads the
  // routing words it rewrites) and the sums it reads; the next lrh_make_timf2 waits for it on the device
  hipEvent_t ev_timf2_done = nullptr, ev_sel_wait = nullptr, ev_sel_wait2 = nullptr; bool timf2_done_valid = false, sel_table_pending = false; hipStream_t sums_stream = nullptr;
  // Two buffers of routing words.  pack_prev_stale: a new table arrived since the last make_timf2 -- it went into the other buffer
  // (pack_new_table swaps the pointers) and d_pack_prev is the table the previous transform was routed with; otherwise that table is
  // d_pack_cur itself and d_pack_prev holds nothing of interest.  (Before: one copy kernel per table, on the path to the next fft2.)
  bool pack_prev_stale = false;
  // pinned staging for mix1 phases
  float *h_ph = nullptr; void *h_ph_dev = nullptr; hipEvent_t ph_ev[LRH_NSTAGE]; int ph_next = 0; size_t ph_stride = 0;
  // mix1 scalars
  lrh_mix1_state ms;
  // masks
  int fft1n_mask, fft1_mask, sumsq_mask, timf2pow_mask, timf2_mask, fft2n_mask, timf3_mask, timf1_bytemask;
  // timers / profiling
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool prof = false, prof_keep_schedule = false; std::map<std::string, ProfEntry> prof_tot; std::vector<ProfPending> prof_pend;
  std::vector<hipEvent_t> ev_pool;
  double host_ms_phases = 0, host_ms_dsp = 0, host_cpu_ms_dsp = 0, host_ms_wait = 0; long host_n_phases = 0, host_n_dsp = 0;   // host CPU time, lrh_profile_get("host:...")
  // Read-backs of a caller that drives the stages from several threads (Linrad's stage threads through integration/hipshim.c): the copy goes to
  // a stream of its own behind an event on the main stream, into a page-locked slot, and the caller waits for it WITHOUT the context's lock --
  // the other stage threads go on enqueueing, and the wait covers what was queued up to the export, not what they add meanwhile.
  bool out_ok = true;               // LRH_OUT_STREAM=0: read-backs on the main stream, waited for under the lock (as before round 5)
  void *out_dst[LRH_NOUT] = {}; size_t out_bytes[LRH_NOUT] = {};
  // Page-locked staging of the library's own (round 6).  Every copy between the device and memory the library does not own -- a caller's table, a
  // std::vector, a stack variable, a numpy array -- goes through it: the HIP runtime never has to pin somebody else's heap pages for a copy.
  // (The round-5 GPU suite died of "Memory access fault by GPU ... on address <a page of the process's malloc heap>" in the one golden case that
  // uploads a small caller table with hipMemcpyAsync)  h_stage: under the context's lock; h_stage_in: the producer's, under mtx_in.
  char *h_stage = nullptr, *h_stage_in = nullptr;
  std::vector<std::pair<char *, size_t>> host_regs;   // spans made page-locked through lrh_host_register: a read-back whose destination lies in one is copied straight there
  int out_ring[LRH_NOUT] = {};       // which ring a slot's copy reads: a stage that rewrites that ring queues its work behind the copy (order_behind_readbacks)
  hipStream_t stream_out = nullptr; void *h_out[LRH_NOUT] = {}; bool out_busy[LRH_NOUT] = {}; hipEvent_t ev_out_src[LRH_NOUT] = {}, ev_out_done[LRH_NOUT] = {};
  // Transforms of the fft1_b workers in stage-call mode (handle >= 1, one call per dispatch of Linrad's wideband thread): a worker's call only
  // NOTES its blocks here (mtx_w; no HIP call); the next reader of fft1_float on the main stream -- lrh_fft1_c of the stage thread, which
  // Linrad wakes once the blocks have been retired in order -- issues everything noted so far as ONE launch per contiguous run, in stream
  // order behind the earlier readers of the ring slots it overwrites.  Measured: with the workers launching themselves (own streams, events
  // both ways) the host's HIP calls bound the drop-in at one block per call: 3 calls per block from each of 3 workers at 12 us apiece under
  // contention (profiles/r05_glue.txt), the device idle 70 % of the time.
  struct ParkedW { int timf1p_ref, fft1_pa, batch; };
  std::vector<ParkedW> wparked; int w_next_nb = -1; std::mutex mtx_evin;
  // lrh_stage_wait: the newest event behind each stage's device work, and whether one has been recorded
  hipEvent_t ev_stage[LRH_STAGE_COUNT] = {}; bool stage_valid[LRH_STAGE_COUNT] = {};
  // ... and a short history of them (round 6): lrh_stage_wait_lag(stage, k) waits for the call k before the newest, so that a stage thread keeps k + 1
  // calls in flight -- one being enqueued while the device works the other off (LRH_STAGE_LAG overrides the caller's k; 0 = the newest, as before)
  hipEvent_t ev_stage_ring[LRH_STAGE_COUNT][4] = {}; unsigned stage_seq[LRH_STAGE_COUNT] = {}; int stage_lag_env = -1;
};

static int fail(lrh_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
  if (c) snprintf(c->err, sizeof c->err, "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
  return code;
}
// A C host (xlinrad64 through integration/hipshim.c, ctypes) must get an error code, never std::terminate: every extern "C" entry point is a
// function-try-block that ends in one of these ("return 0 or a negative code"; lxsys.c:494-505 is where Linrad reports them)
static int lrh_caught(lrh_ctx *c, const char *what) noexcept
{
  if (c) snprintf(c->err, sizeof c->err, "C++ exception inside the library: %s", what ? what : "?");
  fprintf(stderr, "liblinrad_hip: C++ exception caught at the C boundary: %s\n", what ? what : "?");
  return LRH_EINTERNAL;
}
#define LRH_CATCH(c) catch (const std::exception &e_) { return lrh_caught(const_cast<lrh_ctx *>(c), e_.what()); } catch (...) { return lrh_caught(const_cast<lrh_ctx *>(c), "unknown exception"); }
#define LRH_CATCH_NOCTX catch (const std::exception &e_) { return lrh_caught(nullptr, e_.what()); } catch (...) { return lrh_caught(nullptr, "unknown exception"); }
#define LRH_CATCH_OPEN(out) catch (const std::exception &e_) { if (out) *(out) = nullptr; return lrh_caught(nullptr, e_.what()); } catch (...) { if (out) *(out) = nullptr; return lrh_caught(nullptr, "unknown exception"); }
// entry of an API call: the context's lock (see lrh_ctx::mtx) and its device for this host thread
// LRH_CALLPROF=1 (diagnostics): per entry point that takes the context's lock, the calls, the time spent waiting for the lock and the time from there
// to the return, printed by lrh_close (where a stage thread's time goes when several of them share a context: profiles/r06_glue_trace.txt)
static const bool g_callprof = getenv("LRH_CALLPROF") && atoi(getenv("LRH_CALLPROF"));
struct CallProfSite { std::atomic<const char *> name{nullptr}; std::atomic<long long> calls{0}, wait_ns{0}, held_ns{0}; };
static CallProfSite g_callprof_sites[128];
struct CallProfScope {
  const char *fn; std::chrono::steady_clock::time_point t0, t1; bool on;
  explicit CallProfScope(const char *f) : fn(f), on(g_callprof) { if (on) t0 = t1 = std::chrono::steady_clock::now(); }
  void locked() { if (on) t1 = std::chrono::steady_clock::now(); }
  ~CallProfScope() {
    if (!on) return;
    const auto t2 = std::chrono::steady_clock::now();
    for (auto &s : g_callprof_sites) {
      const char *n = s.name.load();
      if (!n) { const char *expect = nullptr; if (!s.name.compare_exchange_strong(expect, fn)) n = expect; else n = fn; }
      if (n == fn) { s.calls++; s.wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count(); s.held_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count(); return; }
    }
  }
};
static void callprof_print()
{
  if (!g_callprof) return;
  for (auto &s : g_callprof_sites) {
    const char *n = s.name.load(); const long long k = s.calls.exchange(0);
    if (!n || !k) continue;
    fprintf(stderr, "LRH_CALLPROF %-28s %8lld calls  lock wait %8.1f us  then %8.1f us  per call\n", n, k, 1e-3 * s.wait_ns.exchange(0) / k, 1e-3 * s.held_ns.exchange(0) / k);
  }
}
#define LRH_LOCK(c) CallProfScope cps_(__func__); std::unique_lock<std::recursive_mutex> lk_; if (c) { lk_ = std::unique_lock<std::recursive_mutex>((c)->mtx); cps_.locked(); hipSetDevice((c)->cfg.device); }
// ... and, for every entry point that may look at or change what the chain has produced, the launches lrh_wideband_dsp still holds
// back from its last round (one-round-late schedule kept across calls, see there) go out first
static int flush_pending(lrh_ctx *c);
static int upload_filtercorr(lrh_ctx *c);
static void join_side_tail(lrh_ctx *c);
static int stage_mark(lrh_ctx *c, int stage);
static int stage_ring_mark(lrh_ctx *c, int stage);
static int join_handles(lrh_ctx *c);
static void pack_new_table(lrh_ctx *c)      // before d_pack_cur is overwritten (by a writer ordered behind the last make_timf2's kernels)
{
  if (!c->pack_prev_stale) std::swap(c->d_pack_cur, c->d_pack_prev);
  c->pack_prev_stale = true;
}
// Read-backs that are still out (lrh_export_begin without its lrh_export_end yet, or another stage thread inside lrh_export with the lock released) read
// their source on the copy stream.  A stage that is about to REWRITE such a ring in place -- the sums of fft1_c, a reused sumsq / waterfall slot,
// fft2_powersum, d_power2 -- first queues the main stream behind those copies (advisor, round 5: torn or newer rows otherwise).  Per ring: a caller that
// collects its read-backs before its stage's next call, as the glue does, never pays for it; holding EVERY later kernel behind every copy cost the
// drop-in 15-30 % (profiles/r06_glue_ab.txt).  Inside lrh_wideband_dsp kernels go to several streams: its entry waits on the host instead.
#define RB(ring) (1u << (ring))
static int order_behind_readbacks(lrh_ctx *c, unsigned ring_mask)
{
  if (!c->stream_out || c->in_dsp) return LRH_OK;
  for (int i = 0; i < LRH_NOUT; i++)
    if (c->out_busy[i] && ((ring_mask >> c->out_ring[i]) & 1u)) { const hipError_t e_ = hipStreamWaitEvent(c->stream, c->ev_out_done[i], 0); if (e_ != hipSuccess) return fail(c, LRH_EDEVICE, "hipStreamWaitEvent(read-back)", e_); }
  return LRH_OK;
}
#define LRH_WRITES(c, mask) do { const int rco_ = order_behind_readbacks(c, (mask)); if (rco_) return rco_; } while (0)
#define LRH_ENTER(c) LRH_LOCK(c); if ((c) && (c)->pend && !(c)->in_dsp) { const int rcf_ = flush_pending(c); if (rcf_) return rcf_; } \
  if ((c) && ((c)->st_pending || (c)->nb_join_pending) && !(c)->in_dsp) join_side_tail(c)
// LRH_HOSTPROF=1 (diagnostics): host time per call site inside lrh_wideband_dsp, printed by lrh_close
struct HostProfSite { double ns = 0; long n = 0; };
static bool g_hostprof = getenv("LRH_HOSTPROF") && atoi(getenv("LRH_HOSTPROF"));
static std::map<std::string, HostProfSite> g_hostprof_sites;
static thread_local int g_hostprof_depth = 0;
struct HostProfTimer {
  const char *what; std::chrono::steady_clock::time_point t0; bool on;
  HostProfTimer(const char *w, bool active) : what(w), on(active && g_hostprof && g_hostprof_depth++ == 0) { if (on) t0 = std::chrono::steady_clock::now(); else if (active && g_hostprof) {} }
  ~HostProfTimer() { if (on) { auto &s = g_hostprof_sites[std::string(what).substr(0, 48)]; s.ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count(); s.n++; g_hostprof_depth--; } }
};
#define HIPCHK(c, call) do { hipError_t e_; { HostProfTimer hp_(#call, (c) && (c)->in_dsp > 0); e_ = (call); if (!hp_.on && (c) && (c)->in_dsp > 0 && g_hostprof) g_hostprof_depth--; } if (e_ != hipSuccess) return fail(c, LRH_EDEVICE, #call, e_); } while (0)

// Device work of a stage function: run now, or (schedule 2 of lrh_wideband_dsp) keep for later.  `body` may use HIPCHK and
// sees the context as `c`; everything else it touches is captured by value.
#define LRH_DEVICE_WORK(c, body)                                                       \
  do {                                                                                 \
    auto op_ = [=](lrh_ctx *c) -> int { body; return LRH_OK; };                        \
    if ((c)->rec) (c)->rec->push_back(op_);                                            \
    else { const int rc_ = op_(c); if (rc_) return rc_; }                              \
  } while (0)

// ------------------------------------------------------------------------------------------------ profiling
struct ProfScope {
  lrh_ctx *c; const char *name; hipEvent_t e0 = nullptr, e1 = nullptr;
  static hipEvent_t get(lrh_ctx *c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
  }
  ProfScope(lrh_ctx *c_, const char *n) : c(c_), name(n) { if (c->prof) { e0 = get(c); e1 = get(c); hipEventRecord(e0, c->cur); } }
  ~ProfScope() { if (c->prof) { hipEventRecord(e1, c->cur); try { c->prof_pend.push_back({name, e0, e1}); } catch (...) {} } }
};
static void prof_collect(lrh_ctx *c)
{
  for (auto &p : c->prof_pend) {
    hipEventSynchronize(p.e1);
    float ms = 0; hipEventElapsedTime(&ms, p.e0, p.e1);
    auto &t = c->prof_tot[p.name]; t.ms += ms; t.n++;
    c->ev_pool.push_back(p.e0); c->ev_pool.push_back(p.e1);
  }
  c->prof_pend.clear();
}

// ------------------------------------------------------------------------------------------------ tables
// make_window (fft0.c:812-921): half window h[0..size/2], sin^n (n 1..7), Gaussian (8), erfc (9); unit mean square
static void half_window(int size, int n, std::vector<float> &h, bool normalise)
{
  h.assign(size / 2 + 1, 0.f);
  double sumsq = 0, x, e1, e2, z = n;
  if (n == 9) {
    e1 = 4.4; e2 = 40.0 / size; if (size < 128) e2 /= 1.5; if (size < 64) e2 /= 1.7;
    for (int i = 0; i <= size / 2; i++) { h[i] = 0.5F * (float)erfc(e1); sumsq += h[i] * h[i]; e1 -= e2; }
  } else if (n == 8) {
    e1 = 0; e2 = 9.8 / size;
    for (int i = size / 2; i >= 0; i--) { h[i] = (float)pow(NATLOG, -e1 * e1); sumsq += h[i] * h[i]; e1 += e2; }
  } else {
    x = 0;
    for (int i = 0; i <= size / 2; i++) { h[i] = (float)pow(sin(x), z); sumsq += h[i] * h[i]; x += PI_L / size; }
  }
  if (normalise) { z = 1 / sqrt(2 * sumsq / size); for (int i = 0; i <= size / 2; i++) h[i] *= (float)z; }
}

// prepare_mixer (buf.c:55-111): the inverted window (make_window mode 3) and the sin^2 / cos^2 crossover functions of a mixer whose window
// is neither none nor sin^2; returns crossover_points
static int prepare_mixer(int Nm, int Im, int Mm, int sp, std::vector<float> &win, std::vector<float> &sin2win, std::vector<float> &cos2win)
{
  std::vector<float> h;
  win.assign(Nm / 2 + 1, 0.f); sin2win.assign(Nm, 0.f); cos2win.assign(Nm, 0.f);
  int X = 0;
  if (sp == 0 || sp == 2) return 0;
  half_window(Nm, sp, h, false);                           // make_window(3,..): inverted, fft0.c:883-891
  win[0] = 1; for (int i = 1; i <= Nm / 2; i++) win[i] = 1 / h[i];
  if (sp == 9) X = Nm / 8;
  else if (sp == 8) X = Nm / 16;
  else {
    unsigned int i = Im / 2;
    const float t1 = win[i];
    while (win[i] < 30 * t1 && i > 0) { i--; X++; }
    if (X > 0.75 * Mm) X = (int)(0.75 * Mm);
    if (X > Im / 2) X = Im / 2;
  }
  float t1 = (float)(0.25 * PI_L / X);
  unsigned int j = (Nm - Mm) / 2, k = j;
  k += X / 2; j -= X / 2;
  for (int i = 0; i < X; i++) {
    cos2win[i] = (float)(win[k] * pow(cos(t1), 2.0));
    sin2win[i] = (float)(win[j] * pow(sin(t1), 2.0));
    k--; j++;
    t1 = (float)(t1 + 0.5 * PI_L / X);
  }
  return X;
}

static float interleave_ratio(int sinpow)   // make_interleave_ratio, buf.c:113-136
{
  if (sinpow == 0) return 0;
  if (sinpow == 9) return 0.625f;
  if (sinpow == 8) return 0.8f;
  return (float)(2 * asin(pow(0.5, 1.0 / sinpow)) / PI_L);
}

static void default_filtercorr(lrh_ctx *c)  // clear_fft1_filtercorr + make_filcorrstart, fft1.c:4653-4724
{
  int N = c->N1;
  float start = 150 * (float)N * (float)pow((double)N, -0.4);
  const bool real = c->cfg.timf1_real_input != 0;
  if (c->cfg.timf1_dword_input) { start *= 4096; start *= real ? 16 : 12; }   /* make_filcorrstart, fft1.c:4656-4663: left-justified int32; real: permute == 2 */
  start = (float)c->cfg.fft1_gain / start;
  c->h_filtercorr.assign(2 * N, 0.f); c->h_desired.assign(N, 1.f);
  for (int i = 0; i < N; i++) c->h_filtercorr[2 * i] = start;
  float t1 = 0.125F * (float)PI_L, t2 = 0, t3;
  int i = 0, k = N - 1;
  while (t2 < 0.5 * PI_L) {
    t3 = (float)(sin((double)t2) * sin((double)t2));        // (double sin as in C: in C++ sin(float) is the float overload, one ulp off in three entries per edge)
    if (!real) { c->h_desired[i] = t3; c->h_filtercorr[2 * i] = t3 * start; }   // fft1.c:4707-4711: the low edge only for I/Q
    c->h_desired[k] = t3; c->h_filtercorr[2 * k] = t3 * start;
    t2 += t1; i++; k--;
  }
}

static void default_yfac(lrh_ctx *c)        // make_wg_yfac, wide_graph.c:955-1001 (second fft, float, 1 channel)
{
  float t1 = (float)(FFT2_WATERFALL_ZERO) / ((float)c->N2 * (float)c->N1);
  t1 /= (float)sqrt((float)(c->cfg.waterfall_avgnum));
  t1 *= (float)(1 << (2 * c->cfg.bckfft_att_n));
  t1 *= (float)(1 + 1 / (0.5 + c->cfg.fft1_sinpow));
  if (c->cfg.blanker_channels == 2) t1 *= 4.0f;            // ui.rx_rf_channels^2, wide_graph.c:985
  c->h_yfac.assign(c->N1, 0.f);
  for (int i = 0; i < c->N1; i++)
    c->h_yfac[i] = (c->h_desired[i] > 0.3162278) ? t1 / (float)pow(c->h_desired[i], 2.0) : t1 * 10;
  c->h_yfac[0] = t1; c->h_yfac[c->N1 - 1] = t1;
}

static void make_twiddles(int N, std::vector<float2> &tw)
{
  tw.resize(N);
  for (int m = 0; m < N; m++) { double a = 2 * PI_L * m / N; tw[m] = make_float2((float)cos(a), (float)-sin(a)); }
}

static void wf_geometry(const lrh_ctx *c, int *hx, int *hp, int *wx, int *wp, int *wfirst)
{
  int r = c->N2 / c->N1; if (r < 1) r = 1;
  int mode = c->cfg.wf_mode;
  if (mode == 1) { *hx = 1; *hp = 1; } else if (mode > 1) { *hx = mode; *hp = 0; } else { *hx = 0; *hp = -mode; }
  if (*hx > 0 && *hx >= r) { *wx = *hx / r; *wp = 0; }
  else { *wx = 0; *wp = *hp > 0 ? *hp * r : r / (*hx > 0 ? *hx : 1); if (*wp < 1) *wp = 1; }
  *wfirst = c->cfg.wf_first_xpoint / r;
}

template <typename T> static int dev_alloc(lrh_ctx *c, T **p, size_t count, bool zero = true)
{
  const size_t spare = guard_mode() == 2 ? 0 : 256;
  hipError_t e = lrh_dev_malloc((void **)p, count * sizeof(T) + spare);
  if (e != hipSuccess) return fail(c, LRH_ENOMEM, "hipMalloc", e);
  // zeroed before anyone can see the pointer: the tables that follow go up with blocking copies on the null stream and the buffer's first kernels may
  // run on any of the context's streams -- none of them is ordered behind a memset that is merely queued on the main stream (round 6: inside lrh_open
  // one wait at the end covered it; the buffers made later -- spur state, blanker tables, scratch grown on demand -- had none)
  if (zero) { e = hipMemsetAsync(*p, 0, count * sizeof(T) + spare, c->stream); if (e == hipSuccess && !c->opening) e = hipStreamSynchronize(c->stream); if (e != hipSuccess) return fail(c, LRH_EDEVICE, "hipMemset", e); }
  return LRH_OK;
}
#define LRH_STAGE_BYTES (1u << 20)
// host -> device and device -> host through the context's page-locked staging buffer, in pieces of LRH_STAGE_BYTES; both return with the data in
// place (the stream is waited for: the buffer is free again and the caller's memory may go away).  A source / destination inside a span the caller has
// page-locked itself (lrh_host_register) is copied directly.
static bool host_span_registered(const lrh_ctx *c, const void *p, size_t bytes)
{
  for (const auto &r : c->host_regs) if ((const char *)p >= r.first && (const char *)p + bytes <= r.first + r.second) return true;
  return false;
}
static hipError_t stage_h2d(lrh_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st, char *stage = nullptr)
{
  if (!bytes) return hipSuccess;
  if (!stage) stage = c->h_stage;
  if (!stage || host_span_registered(c, src, bytes)) { const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); return e != hipSuccess ? e : hipStreamSynchronize(st); }
  for (size_t off = 0; off < bytes; off += LRH_STAGE_BYTES) {
    const size_t n = bytes - off < LRH_STAGE_BYTES ? bytes - off : LRH_STAGE_BYTES;
    memcpy(stage, (const char *)src + off, n);
    hipError_t e = hipMemcpyAsync((char *)dst + off, stage, n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
static hipError_t stage_d2h(lrh_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st)
{
  if (!bytes) return hipSuccess;
  if (!c->h_stage || host_span_registered(c, dst, bytes)) { const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st); return e != hipSuccess ? e : hipStreamSynchronize(st); }
  for (size_t off = 0; off < bytes; off += LRH_STAGE_BYTES) {
    const size_t n = bytes - off < LRH_STAGE_BYTES ? bytes - off : LRH_STAGE_BYTES;
    hipError_t e = hipMemcpyAsync(c->h_stage, (const char *)src + off, n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    memcpy((char *)dst + off, c->h_stage, n);
  }
  return hipSuccess;
}
template <typename T> static int upload(lrh_ctx *c, T *dst, const T *src, size_t count)
{
  HIPCHK(c, stage_h2d(c, dst, src, count * sizeof(T), c->stream));
  return LRH_OK;
}
static int ispow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ------------------------------------------------------------------------------------------------ API
extern "C" {

int lrh_abi_version(void) { return LRH_ABI_VERSION; }
// what the C boundary makes of an exception (tests/test_abi_cpu.py; needs no device): the same function-try-block every entry point ends in
int lrh_selftest_exception(int kind)
try {
  if (kind == 0) throw std::bad_alloc();
  if (kind == 1) { std::vector<int> v; return v.at(3); }      // std::out_of_range out of the standard library
  if (kind == 2) throw 42;                                   // not derived from std::exception
  return LRH_OK;
}
LRH_CATCH_NOCTX
size_t lrh_sizeof(int which)
try {
  static const size_t sz[] = { sizeof(lrh_config), sizeof(lrh_ptrs), sizeof(lrh_blanker_state), sizeof(lrh_blanker_tables), sizeof(lrh_mix1_state),
                               sizeof(lrh_sellim), sizeof(lrh_spur), sizeof(lrh_afc), sizeof(lrh_synth) };
  return which >= 0 && which < (int)(sizeof sz / sizeof sz[0]) ? sz[which] : 0;
}
LRH_CATCH_NOCTX

int lrh_config_defaults(lrh_config *c, int fft1_n, int fft2_n)
try {
  if (!c) return LRH_EINVAL;
  memset(c, 0, sizeof *c);
  int N1 = 1 << fft1_n, N2 = 1 << fft2_n, NM = N1 > N2 ? N1 : N2;
  c->struct_size = (int)sizeof *c; c->device = 0; c->rx_rf_channels = 1;
  c->fft1_n = fft1_n; c->fft1_sinpow = 2; c->fft1_gain = 2000; c->fft1_direction = 1;   /* uivar.c:371 */
  c->fft_avg1num = 5; c->fft_avg2num = 4; c->timf1_bytes = 64 * N1 * 4; c->max_fft1n = 32;
  c->fft1_sumsq_bufsize = 8 * N1; c->wg_xpoints = N1 - 1; c->slowsum_fresh_recalc = 2;
  c->bckfft_att_n = 6; c->timf2pow_size = 8 * NM;
  c->stupid_bln_mode = 1; c->stupid_bln_factor = 5; c->blnfit_range = 48; c->blanker_pulsewidth = 0;
  c->timf2_noise_floor_avgnum = 32; c->blanker_info_update_interval = 4; c->blanker_min_points = N2 / 3; c->timf2_noise_floor = 200;
  c->fft2_n = fft2_n; c->fft2_sinpow = 2; c->max_fft2n = 4; c->waterfall_avgnum = 2; c->wf_first_xpoint = 0;
  c->wf_xpixels = N2 < 1024 ? N2 : 1024; c->wf_mode = 1; c->wf_lines = 8;
  c->mix1_bandwidth_reduction_n = 6; c->timf3_size = 32 * ((N2 >> 6) > 8 ? (N2 >> 6) : 8);
  c->fftx_points_per_hz = 1.0f; c->mix1_lowest_fq = 0; c->mix1_highest_fq = (float)N2; c->max_batch = 16;
  c->second_fft_enable = 1; c->timf2_blockpower_block = 0; c->timf2_blockpower_size = 1024;
  c->fft3_n = 0; c->fft3_sinpow = 2; c->mix2_n = 0; c->max_fft3n = 8; c->baseband_size = 4096;
  return LRH_OK;
}
LRH_CATCH_NOCTX

void lrh_close(lrh_ctx *c)
try {
  if (!c) return;
  callprof_print();
  if (g_hostprof && !g_hostprof_sites.empty()) {
    std::vector<std::pair<std::string, HostProfSite>> v(g_hostprof_sites.begin(), g_hostprof_sites.end());
    std::sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return a.second.ns > b.second.ns; });
    double tot = 0; long n = 0; for (auto &e : v) { tot += e.second.ns; n += e.second.n; }
    fprintf(stderr, "LRH_HOSTPROF: %ld calls, %.3f ms inside HIP calls of lrh_wideband_dsp (%ld calls of it, %.3f ms wall, %.3f ms cpu, of which %.3f ms in the staging wait, %.3f ms phase tables)\n", n, tot * 1e-6, c->host_n_dsp, c->host_ms_dsp, c->host_cpu_ms_dsp, c->host_cpu_ms_wait, c->host_ms_phases);
    for (size_t i = 0; i < v.size() && i < 40; i++) fprintf(stderr, "  %-48s %7ld x %7.2f us = %8.3f ms\n", v[i].first.c_str(), v[i].second.n, v[i].second.ns * 1e-3 / v[i].second.n, v[i].second.ns * 1e-6);
    g_hostprof_sites.clear();
  }
  if (alloc_log()) alloc_note('C', c, 0);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->stream2) { hipStreamSynchronize(c->stream2); hipStreamDestroy(c->stream2); }
  if (c->stream_sel) { hipStreamSynchronize(c->stream_sel); hipStreamDestroy(c->stream_sel); }
  if (c->stream3) { hipStreamSynchronize(c->stream3); hipStreamDestroy(c->stream3); }
  if (c->stream_in) { hipStreamSynchronize(c->stream_in); hipStreamDestroy(c->stream_in); }
  for (int h = 0; h < LRH_MAX_HANDLES; h++) { if (c->hstream[h]) { hipStreamSynchronize(c->hstream[h]); hipStreamDestroy(c->hstream[h]); } if (c->hev[h]) hipEventDestroy(c->hev[h]); }
  if (c->hev_start) hipEventDestroy(c->hev_start);
  if (c->ev_in) hipEventDestroy(c->ev_in);
  for (auto &m : c->in_marks) if (m.ev) hipEventDestroy(m.ev);
  if (c->ev_fft1_read) hipEventDestroy(c->ev_fft1_read);
  if (c->ev_in_guard) hipEventDestroy(c->ev_in_guard);
  if (c->stream_nb) { hipStreamSynchronize(c->stream_nb); hipStreamDestroy(c->stream_nb); }
  if (c->stream_out) { hipStreamSynchronize(c->stream_out); hipStreamDestroy(c->stream_out); }
  if (c->stream_f2) { hipStreamSynchronize(c->stream_f2); hipStreamDestroy(c->stream_f2); }
  if (c->stream_ss) { hipStreamSynchronize(c->stream_ss); hipStreamDestroy(c->stream_ss); }
  if (c->ev_ss_in) hipEventDestroy(c->ev_ss_in);
  if (c->ev_ss_done) hipEventDestroy(c->ev_ss_done);
  for (int i = 0; i < 3; i++) { if (c->ev_f2c[i]) hipEventDestroy(c->ev_f2c[i]); if (c->ev_f2r[i]) hipEventDestroy(c->ev_f2r[i]); }
  for (int i = 0; i < LRH_NOUT; i++) { if (c->h_out[i]) lrh_host_free(c->h_out[i]); if (c->ev_out_src[i]) hipEventDestroy(c->ev_out_src[i]); if (c->ev_out_done[i]) hipEventDestroy(c->ev_out_done[i]); }
  for (int i = 0; i < LRH_STAGE_COUNT; i++) if (c->ev_stage[i]) hipEventDestroy(c->ev_stage[i]);
  for (int i = 0; i < LRH_STAGE_COUNT; i++) for (int k = 0; k < 4; k++) if (c->ev_stage_ring[i][k]) hipEventDestroy(c->ev_stage_ring[i][k]);
  for (hipEvent_t ev : { c->ev_st[0], c->ev_st[1], c->ev_blank2[0], c->ev_blank2[1], c->ev_f2done, c->ev_nb_ring[0], c->ev_nb_ring[1], c->ev_nb_ring[2], c->ev_nb_ring[3], c->ev_timf2_done, c->ev_sel_wait, c->ev_sel_wait2, c->ev_fft1, c->ev_timf2, c->ev_blank, c->ev_fft2, c->ev_side, c->ev_ps2, c->ev_sumsq[0], c->ev_sumsq[1], c->ev_tail, c->ev_timf2b }) if (ev) hipEventDestroy(ev);
  void *dev[] = { c->d_ss_sum, c->d_ss_spec_base, c->d_ss_min, c->d_ss_out, c->d_bbfir, c->d_mix2win, c->d_sin2win2, c->d_cos2win2, c->d_filtercorr_v, c->d_mixwin, c->d_sin2win, c->d_cos2win, c->d_window1, c->d_invwin1, c->d_window2, c->d_fqwin, c->d_yfac, c->d_filtercorr, c->d_tw1, c->d_tw2, c->d_twm,
                  c->d_pack_cur, c->d_pack_prev, c->d_wf_itab, c->d_timf1, c->d_fft1, c->d_sumsq, c->d_slowsum, c->d_timf2w, c->d_timf2s, c->d_pwr,
                  c->d_blnbits, c->d_fft2, c->d_power2, c->d_powersum2, c->d_powersum2_alt, c->d_wf_scratch, c->d_waterf, c->d_timf3, c->d_mix_scratch,
                  c->d_ph, c->d_bst, c->d_partials, c->d_bln_tiles, c->d_bln_counts, c->d_bln_wbusy, c->d_bln_wstate, c->d_bt_refpulse, c->d_bt_phasefunc, c->d_bt_pulindex, c->d_bln_flag, c->d_bln_cand, c->d_sel_ftmp, c->d_sel_reg, c->d_sel_desired, c->d_sel_bigb, c->d_sel_bigg, c->d_clv_amp, c->d_clv_dbg, c->d_clv_start, c->d_clv_ext, c->d_clv_ctl, c->d_clv_bk_pos, c->d_clv_logged, c->d_clv_bk_pwr, c->d_clv_bk_tf, c->d_clv_bk_pwo, c->d_clv_bk_ty, c->d_liminfo, c->d_old_liminfo, c->d_sel_tmp, c->d_sel_wait, c->d_sel_st, c->d_ss_part, c->d_pwr_sum, c->d_xbuf, c->d_xstat, c->d_xweak, c->d_tf_partner, c->d_xspec, c->d_corrsum, c->d_slowcorr, c->d_slowcorr_tot, c->d_xbins, c->d_xypower, c->d_xysum, c->d_xysum_alt, c->d_xpol, c->d_tw2a, c->d_tw2b, c->d_fft2_scratch, c->d_tw1a, c->d_tw1b, c->d_timf2_scratch, c->d_fft1_scratch[0], c->d_fft1_scratch[1], c->d_fft1_scratch[2], c->d_fft1_scratch[3],
                  c->d_fft1_scratch[4], c->d_fft1_scratch[5], c->d_fft1_scratch[6], c->d_blockpower,
                  c->d_window3, c->d_bgfilt, c->d_tw3, c->d_twm2, c->d_fft3, c->d_baseb, c->d_mix2_scratch };
  for (auto &r : c->host_regs) hipHostUnregister(r.first);   // spans the caller left page-locked (lrh_host_register without its lrh_host_unregister): the context is their owner of record
  c->host_regs.clear();
  for (void *p : dev) if (p) lrh_dev_free(p);
  if (c->h_ph) lrh_host_free(c->h_ph);
  if (c->h_stage) lrh_host_free(c->h_stage);
  if (c->h_stage_in) lrh_host_free(c->h_stage_in);
  if (c->h_sel_low) lrh_host_free(c->h_sel_low);
  if (c->h_clv_out) lrh_host_free(c->h_clv_out);
  if (c->ev_clv) hipEventDestroy(c->ev_clv);
  if (c->ev_amp) hipEventDestroy(c->ev_amp);
  if (c->ev_sel) hipEventDestroy(c->ev_sel);
  for (hipEvent_t e : c->ev_sel_slot) if (e) hipEventDestroy(e);
  if (c->d_pack18) lrh_dev_free(c->d_pack18);
  if (c->d_stamps) lrh_dev_free(c->d_stamps);
  for (void *q_ : { (void *)c->d_spurs, (void *)c->d_spur_table, (void *)c->d_spur_signal, (void *)c->d_spur_touched, (void *)c->d_spur_spectra, (void *)c->d_spur_ind }) if (q_) lrh_dev_free(q_);
  if (c->d_net) lrh_dev_free(c->d_net);
  if (c->d_fft1net) lrh_dev_free(c->d_fft1net);
  if (c->d_foldcorr) lrh_dev_free(c->d_foldcorr);
  if (c->d_unitcorr) lrh_dev_free(c->d_unitcorr);
  for (int i = 0; i < LRH_NSTAGE; i++) if (c->ph_ev[i]) hipEventDestroy(c->ph_ev[i]);
  for (auto &p : c->prof_pend) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
  for (auto e : c->ev_pool) hipEventDestroy(e);
  if (c->t0) hipEventDestroy(c->t0);
  if (c->t1) hipEventDestroy(c->t1);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}
catch (...) { lrh_caught(nullptr, "exception inside lrh_close"); }

const char *lrh_last_error(const lrh_ctx *c) { return c ? c->err : "null context"; }

int lrh_open(const lrh_config *cfg, lrh_ctx **out)
try {
  if (!cfg || !out || cfg->struct_size != (int)sizeof(lrh_config)) return LRH_EINVAL;
  *out = nullptr;
  if (cfg->rx_rf_channels != 1) return LRH_EINVAL;                      // channels shard one per context / GPU
  if (cfg->timf1_real_input && (cfg->timf1_frame_channels > 2 || cfg->sample_shift != 0)) return LRH_EINVAL;   // fft1_reherm_dit_one / _two: one or two real channels per frame
  if (cfg->timf1_frame_channels > 1 && (!ispow2(cfg->timf1_frame_channels) || cfg->timf1_channel_index < 0 || cfg->timf1_channel_index >= cfg->timf1_frame_channels)) return LRH_EINVAL;
  if (cfg->fft1_n < 6 || cfg->fft1_n > 16 || cfg->fft2_n < 6 || cfg->fft2_n > 18) return LRH_EINVAL;   // fft2 > 16384, fft1 >= 32768: four-step
  if (cfg->fft1_n == 16 && cfg->second_fft_enable) return LRH_EINVAL;                                  // 65536 only without the second fft (buf.c:335, fft0.c:1162-1169)
  // fft1_size 32768 (buf.c:335): I/Q samples (int16 / int32) through a sin^2 window, the second fft's configuration; the variants
  // that only exist as single-workgroup kernels (real input, I/Q skew, other windows) are refused
  if (cfg->fft1_n >= 15 && (cfg->timf1_real_input || cfg->sample_shift || cfg->fft1_sinpow != 2)) return LRH_EINVAL;
  if (!ispow2(cfg->timf1_bytes) || !ispow2(cfg->max_fft1n) || !ispow2(cfg->fft1_sumsq_bufsize) || !ispow2(cfg->timf2pow_size) ||
      !ispow2(cfg->max_fft2n) || !ispow2(cfg->timf3_size) || (cfg->timf2_blockpower_block > 0 && !ispow2(cfg->timf2_blockpower_size)) || cfg->max_batch < 1 || cfg->wf_xpixels < 1 || cfg->wf_lines < 1) return LRH_EINVAL;
  lrh_ctx *c = new lrh_ctx();
  if (alloc_log()) alloc_note('O', c, (size_t)cfg->fft1_n * 100 + cfg->fft2_n);
  memset(c->ph_ev, 0, sizeof c->ph_ev);
  c->cfg = *cfg; c->opening = true;
  const int N1 = c->N1 = 1 << cfg->fft1_n, N2 = c->N2 = 1 << cfg->fft2_n;
  c->I1 = (int)(1 + interleave_ratio(cfg->fft1_sinpow) * N1); c->I1 &= 0xfffe;                           // buf.c:303-304
  if (cfg->second_fft_enable) {
    c->mix1_n = cfg->fft2_n - cfg->mix1_bandwidth_reduction_n; if (c->mix1_n < 3) c->mix1_n = 3;          // buf.c:432-434
    c->Nm = 1 << c->mix1_n;
    c->Im = (int)(interleave_ratio(cfg->fft2_sinpow) * c->Nm); c->Im &= 0xfffffffe; c->Mm = c->Nm - c->Im; // buf.c:451-452
    c->I2 = c->Im * (N2 / c->Nm); c->M2 = N2 - c->I2;                                                       // buf.c:453-455
  } else {                                   // buf.c:315-327: mix1 sized from fft1, fft1 interleave re-derived from it
    c->mix1_n = cfg->fft1_n - cfg->mix1_bandwidth_reduction_n; if (c->mix1_n < 3) c->mix1_n = 3;
    c->Nm = 1 << c->mix1_n;
    c->Im = (int)(interleave_ratio(cfg->fft1_sinpow) * c->Nm); c->Im &= 0xfffffffe; c->Mm = c->Nm - c->Im;
    c->I1 = c->Im * (N1 / c->Nm);
    c->I2 = 0; c->M2 = N2;
  }
  c->M1 = N1 - c->I1;
  c->mix_cap = cfg->max_fft2n > cfg->max_batch ? cfg->max_fft2n : cfg->max_batch;
  if (cfg->fft3_n > 0) {                                        // baseb_graph.c:636-645
    if (cfg->fft3_n < 6 || cfg->fft3_n > 14 || cfg->mix2_n < 3 || cfg->mix2_n > cfg->fft3_n || !ispow2(cfg->max_fft3n) || !ispow2(cfg->baseband_size)) { delete c; return LRH_EINVAL; }
    c->N3 = 1 << cfg->fft3_n; c->Nm2 = 1 << cfg->mix2_n;
    c->Im2 = (int)(interleave_ratio(cfg->fft3_sinpow) * c->Nm2); c->Im2 &= 0xfffffffe; c->Mm2 = c->Nm2 - c->Im2;
    c->I3 = c->Im2 * (c->N3 / c->Nm2); c->M3 = c->N3 - c->I3;
    if (cfg->timf3_size < 4 * c->N3 || cfg->baseband_size < 4 * c->Nm2) { delete c; return LRH_EINVAL; }
  }
  c->timf2_mode = c->I1 == 0 ? 0 : (c->I1 == N1 / 2 ? 1 : 2);
  if (const char *e = getenv("LRH_XCD_MASK")) c->xcd_mask = atoi(e);
  // one workgroup per waterfall averaging group keeps the power sums in registers; long groups would starve the chip
  c->fft2_fused = cfg->second_fft_enable && (cfg->fft2_n <= LRH_FFT2_FUSED_MAXLOG || cfg->fft2_n > 14) && cfg->waterfall_avgnum >= 1 && cfg->waterfall_avgnum <= 16;
  if (const char *e = getenv("LRH_FFT2_FUSED")) c->fft2_fused = c->fft2_fused && atoi(e) != 0;
  bool bad = cfg->fft1_sumsq_bufsize < 2 * N1 || cfg->fft1_sumsq_bufsize < (cfg->fft_avg2num + 1) * N1 ||
             cfg->timf2pow_size < 2 * N1 || cfg->timf2pow_size < 2 * N2 || cfg->max_fft1n < 2 * cfg->max_batch ||
             cfg->timf3_size < 4 * c->Nm || cfg->timf1_bytes < 8 * N1 ||
             (cfg->second_fft_enable && (size_t)cfg->max_batch * c->M1 + N1 > (size_t)cfg->timf2pow_size);   // (nothing goes into timf2 with the second fft off)
  if (bad) { delete c; return LRH_EINVAL; }
  c->fft1n_mask = cfg->max_fft1n - 1; c->fft1_mask = cfg->max_fft1n * 2 * N1 - 1; c->sumsq_mask = cfg->fft1_sumsq_bufsize - 1;
  c->timf2pow_mask = cfg->timf2pow_size - 1; c->timf2_mask = 4 * cfg->timf2pow_size - 1; c->fft2n_mask = cfg->max_fft2n - 1;
  c->timf3_mask = cfg->timf3_size - 1; c->timf1_bytemask = cfg->timf1_bytes - 1;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= cfg->device) { int rc = fail(c, LRH_EDEVICE, "no HIP device", e); delete c; return rc; }
  if ((e = hipSetDevice(cfg->device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking)) != hipSuccess) { lrh_close(c); return LRH_EDEVICE; }
  c->cur = c->stream;
  if (hipStreamCreateWithFlags(&c->stream_nb, hipStreamNonBlocking) != hipSuccess) { lrh_close(c); return LRH_EDEVICE; }
  if (const char *e_ = getenv("LRH_NARROW_STREAM")) c->nb_split = atoi(e_) != 0;
  if (const char *e_ = getenv("LRH_SIDE_TAIL")) c->st_on = atoi(e_) != 0;
  for (hipEvent_t *ev : { &c->ev_st[0], &c->ev_st[1], &c->ev_blank2[0], &c->ev_blank2[1], &c->ev_f2done, &c->ev_nb_ring[0], &c->ev_nb_ring[1], &c->ev_nb_ring[2], &c->ev_nb_ring[3], &c->ev_timf2_done, &c->ev_sel_wait, &c->ev_fft1, &c->ev_timf2, &c->ev_blank, &c->ev_fft2, &c->ev_side, &c->ev_ps2, &c->ev_sumsq[0], &c->ev_sumsq[1], &c->ev_tail, &c->ev_timf2b }) hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (const char *e2 = getenv("LRH_PIPELINE")) { c->pipeline = atoi(e2); c->pipeline_forced = true; }
  if (const char *e3 = getenv("LRH_EARLY_UPLOAD")) c->early_upload = atoi(e3) != 0;
  if (const char *e4 = getenv("LRH_FUSE_SUMSQ")) c->fuse_sumsq = atoi(e4) != 0;
  if (const char *e9v = getenv("LRH_FFT1V")) c->fuse_v = atoi(e9v) != 0;
  if (const char *e9r = getenv("LRH_FUSE_REAL")) c->fuse_real = atoi(e9r) != 0;
  if (const char *e9s = getenv("LRH_FFT2_SPAN")) c->fft2_span = atoi(e9s);
  if (const char *e9 = getenv("LRH_FUSE_FFT1")) { c->fuse_fft1 = atoi(e9) != 0; c->fuse_fft1_forced = c->fuse_fft1; }   // 0: k_fft1 + k_timf2 also where k_fft1w would run; 1: k_fft1w also for rounds of a few blocks
  c->sums_on_main = cfg->fft2_n <= 14;
  if (const char *e7 = getenv("LRH_SUMS_MAIN")) c->sums_on_main = atoi(e7) != 0;
  if (const char *e6 = getenv("LRH_SPARE_CUS")) { c->spare_cus = atoi(e6); if (c->spare_cus < 0 || c->spare_cus > 128) c->spare_cus = 0; }
  if (const char *e8 = getenv("LRH_PERSIST")) c->persist = atoi(e8) != 0;
  if (const char *e8 = getenv("LRH_OUT_STREAM")) c->out_ok = atoi(e8) != 0;
  if (const char *e9 = getenv("LRH_IN_MERGE_KB")) { const long v = atol(e9); c->in_merge_bytes = v <= 0 ? 0 : (size_t)(v > (1 << 20) ? (1 << 20) : v) << 10; }
  if (const char *e8 = getenv("LRH_STAGE_LAG")) { c->stage_lag_env = atoi(e8); if (c->stage_lag_env > 3) c->stage_lag_env = 3; }
  if (const char *e8 = getenv("LRH_WORKER_FAST")) c->worker_fast = atoi(e8) != 0;
  if (const char *e5 = getenv("LRH_CLEVER_SERIAL")) c->clever_force_serial = atoi(e5) != 0;
  if (const char *e6 = getenv("LRH_CLEVER_FIRST")) c->clv_first = atoi(e6);
  if (const char *e7 = getenv("LRH_CLEVER_SPLIT")) c->clv_split = atoi(e7);   // tests: the one-wave replay of the linear blanker
  if (const char *e5 = getenv("LRH_STAMP")) c->dbg_stamp = atoi(e5);
  if (const char *e6 = getenv("LRH_BLN_DEBUG")) c->dbg_bln = atoi(e6);
  if (const char *e7 = getenv("LRH_FFT2_RUN")) c->env_fft2_run = atoi(e7);
  if (const char *e8 = getenv("LRH_FFT2_COLS_RUN")) c->env_fft2_cols_run = atoi(e8);
  hipEventCreate(&c->t0); hipEventCreate(&c->t1);
  hipStreamCreateWithFlags(&c->stream_in, hipStreamNonBlocking);
  hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming); hipEventCreateWithFlags(&c->ev_fft1_read, hipEventDisableTiming);
  int rc = LRH_OK;
#define A(call) do { if (rc == LRH_OK) rc = (call); } while (0)
  // ---- tables
  const bool real1 = cfg->timf1_real_input != 0;
  std::vector<float> h, win1(real1 ? 2 * N1 : N1, 1.0f), inv1(N1, 1.0f);
  c->h_window1_ref.assign(N1 + 1, 0.f); c->h_invwin1_ref.assign(N1 / 2 + 1, 0.f);
  if (real1 && cfg->fft1_sinpow) {          // make_window(2,..): half window of the 2*N1-point transform, win[0..N1]; sample ia
    half_window(2 * N1, cfg->fft1_sinpow, h, true);      // and sample 2*N1-1-ia both take win[ia] (fft1_re.c:47-57)
    for (int i = 0; i < N1; i++) { win1[i] = h[i]; win1[2 * N1 - 1 - i] = h[i]; }
    for (int i = 0; i <= N1; i++) c->h_window1_ref[i] = h[i];
  }
  if (cfg->fft1_sinpow) {
    half_window(N1, cfg->fft1_sinpow, h, true);
    if (!real1) {
    for (int i = 0; i <= N1 / 2; i++) win1[i] = h[i];
    for (int i = N1 / 2 + 1; i < N1; i++) win1[i] = h[N1 - i];
    for (int i = 0; i < N1 / 2; i++) { c->h_window1_ref[2 * i] = h[i]; c->h_window1_ref[2 * i + 1] = h[N1 / 2 - i]; }   // fft0.c:907-920
    }
    if (cfg->fft1_sinpow != 2) {                                       // make_window(3,..) fft0.c:883-891
      half_window(N1, cfg->fft1_sinpow, h, false);
      c->h_invwin1_ref[0] = 1; for (int i = 1; i <= N1 / 2; i++) c->h_invwin1_ref[i] = 1 / h[i];
      for (int i = 0; i <= N1 / 2; i++) inv1[i] = c->h_invwin1_ref[i];
      for (int i = N1 / 2 + 1; i < N1; i++) inv1[i] = c->h_invwin1_ref[N1 - i];
    }
  }
  c->h_window2.assign(N2, 1.0f);
  if (cfg->fft2_sinpow) {
    half_window(N2, cfg->fft2_sinpow, h, true);
    for (int i = 0; i <= N2 / 2; i++) c->h_window2[i] = h[i];
    for (int i = N2 / 2 + 1; i < N2; i++) c->h_window2[i] = h[N2 - i];
  }
  // prepare_mixer(&mix1, ..) (buf.c:1290) and, with fft3 configured, prepare_mixer(&mix2, THIRD_FFT_SINPOW) (baseb_graph.c:899)
  c->Xm = prepare_mixer(c->Nm, c->Im, c->Mm, cfg->second_fft_enable ? cfg->fft2_sinpow : cfg->fft1_sinpow, c->h_mixwin, c->h_sin2win, c->h_cos2win);
  if (cfg->fft3_n > 0) c->Xm2 = prepare_mixer(c->Nm2, c->Im2, c->Mm2, cfg->fft3_sinpow, c->h_mix2win, c->h_sin2win2, c->h_cos2win2);
  c->h_fqwin.assign(c->Nm / 2 + 1, 0.f);
  { double e1 = 3.2, e2 = 13.0 / c->Nm; for (int i = 0; i <= c->Nm / 2; i++) { c->h_fqwin[i] = 0.5F * (float)erfc(e1); e1 -= e2; } }   // fft0.c:818-827
  default_filtercorr(c); default_yfac(c);
  std::vector<float2> tw1, tw2, twm; make_twiddles(N1, tw1); make_twiddles(N2, tw2); make_twiddles(c->Nm, twm);
  // waterfall yfac index per pixel / data point, accumulated in float like fft2.c:713-729
  int hx, hp, wx, wp, wfirst; wf_geometry(c, &hx, &hp, &wx, &wp, &wfirst);
  std::vector<int> itab(cfg->wf_xpixels + 4);
  { float a2 = 1, a3; if (wx > 0) a2 = wx; else a2 = 1. / wp; a3 = wfirst + 0.5 * a2;
    for (size_t i = 0; i < itab.size(); i++) { int t = a3; if (t < 0) t = 0; if (t > N1 - 1) t = N1 - 1; itab[i] = t; a3 += a2; } }
  A(dev_alloc(c, &c->d_window1, real1 ? 2 * N1 : N1)); A(dev_alloc(c, &c->d_invwin1, N1)); A(dev_alloc(c, &c->d_window2, N2));
  A(dev_alloc(c, &c->d_fqwin, c->Nm / 2 + 1)); A(dev_alloc(c, &c->d_yfac, N1)); A(dev_alloc(c, &c->d_filtercorr, N1));
  if (cfg->fft1_n >= 12 && cfg->fft1_n <= 14) A(dev_alloc(c, &c->d_filtercorr_v, N1));
  A(dev_alloc(c, &c->d_mixwin, c->Nm / 2 + 1)); A(dev_alloc(c, &c->d_sin2win, c->Nm)); A(dev_alloc(c, &c->d_cos2win, c->Nm));
  A(dev_alloc(c, &c->d_tw1, N1)); A(dev_alloc(c, &c->d_tw2, N2)); A(dev_alloc(c, &c->d_twm, c->Nm));
  const int fft2_la = cfg->fft2_n - cfg->fft2_n / 2, fft2_lb = cfg->fft2_n / 2;      // four-step split NA x NB
  std::vector<float2> tw2a, tw2b;
  if (cfg->fft2_n > 14) {
    make_twiddles(1 << fft2_la, tw2a); make_twiddles(1 << fft2_lb, tw2b);
    A(dev_alloc(c, &c->d_tw2a, tw2a.size())); A(dev_alloc(c, &c->d_tw2b, tw2b.size()));
    A(dev_alloc(c, &c->d_fft2_scratch, (size_t)cfg->max_fft2n * N2, false));
  }
  std::vector<float2> tw1a, tw1b;
  c->fft1_big = cfg->fft1_n >= 15;
  if (c->fft1_big) {
    make_twiddles(256, tw1a); make_twiddles(N1 / 256, tw1b);
    A(dev_alloc(c, &c->d_tw1a, tw1a.size())); A(dev_alloc(c, &c->d_tw1b, tw1b.size()));
    if (cfg->fft1_n != 15) c->fuse_sumsq = false;   // 65536 (second fft off): fft1_c's sums stay a separate pass (k_sumsq); 32768: inside k_fft1r_t2c
  }
  A(dev_alloc(c, &c->d_pack_cur, N1)); A(dev_alloc(c, &c->d_pack_prev, N1));
  A(dev_alloc(c, &c->d_liminfo, N1)); A(dev_alloc(c, &c->d_old_liminfo, N1)); A(dev_alloc(c, &c->d_sel_tmp, N1)); A(dev_alloc(c, &c->d_sel_wait, N1)); A(dev_alloc(c, &c->d_sel_st, 1)); A(dev_alloc(c, &c->d_wf_itab, itab.size()));
  // ---- rings
  A(dev_alloc(c, &c->d_timf1, cfg->timf1_bytes / 4)); A(dev_alloc(c, &c->d_fft1, (size_t)cfg->max_fft1n * N1));
  A(dev_alloc(c, &c->d_sumsq, cfg->fft1_sumsq_bufsize)); A(dev_alloc(c, &c->d_slowsum, N1));
  A(dev_alloc(c, &c->d_timf2w, cfg->timf2pow_size)); A(dev_alloc(c, &c->d_timf2s, cfg->timf2pow_size)); A(dev_alloc(c, &c->d_pwr, cfg->timf2pow_size));
  A(dev_alloc(c, &c->d_blnbits, cfg->timf2pow_size / 32 + 64));
  A(dev_alloc(c, &c->d_fft2, (size_t)cfg->max_fft2n * N2)); A(dev_alloc(c, &c->d_power2, (size_t)cfg->max_fft2n * N2));
  A(dev_alloc(c, &c->d_powersum2, N2)); A(dev_alloc(c, &c->d_powersum2_alt, N2)); A(dev_alloc(c, &c->d_wf_scratch, (size_t)(cfg->max_fft2n + 1) * N2));
  A(dev_alloc(c, &c->d_waterf, (size_t)cfg->wf_lines * cfg->wf_xpixels + 64));
  A(dev_alloc(c, &c->d_timf3, cfg->timf3_size / 2 + c->Nm)); A(dev_alloc(c, &c->d_mix_scratch, (size_t)c->mix_cap * c->Nm));
  A(dev_alloc(c, &c->d_blockpower, cfg->timf2_blockpower_size > 0 ? cfg->timf2_blockpower_size : 1));
  std::vector<float> win3; std::vector<float2> tw3, twm2;
  if (c->N3) {
    win3.assign(c->N3, 1.0f); c->h_window3_ref.assign(c->N3, 0.f);
    if (cfg->fft3_sinpow) {
      half_window(c->N3, cfg->fft3_sinpow, h, true);
      for (int i = 0; i <= c->N3 / 2; i++) win3[i] = h[i];
      for (int i = c->N3 / 2 + 1; i < c->N3; i++) win3[i] = h[c->N3 - i];
      for (int i = 0; i < c->N3 / 2; i++) { c->h_window3_ref[2 * i] = h[i]; c->h_window3_ref[2 * i + 1] = h[c->N3 / 2 - i]; }
    }
    make_twiddles(c->N3, tw3); make_twiddles(c->Nm2, twm2);
    A(dev_alloc(c, &c->d_window3, c->N3)); A(dev_alloc(c, &c->d_bgfilt, c->N3)); A(dev_alloc(c, &c->d_tw3, c->N3)); A(dev_alloc(c, &c->d_twm2, c->Nm2));
    A(dev_alloc(c, &c->d_fft3, (size_t)cfg->max_fft3n * c->N3)); A(dev_alloc(c, &c->d_baseb, (size_t)cfg->baseband_size + 2 * c->Nm2));
    A(dev_alloc(c, &c->d_mix2_scratch, (size_t)cfg->max_fft3n * c->Nm2));
    A(dev_alloc(c, &c->d_xpol, (size_t)2 * cfg->max_fft3n * c->Nm2));
    A(dev_alloc(c, &c->d_mix2win, c->Nm2 / 2 + 1)); A(dev_alloc(c, &c->d_sin2win2, c->Nm2)); A(dev_alloc(c, &c->d_cos2win2, c->Nm2));
  }
  c->ph_stride = (size_t)2 * c->mix_cap * c->Nm;
  A(dev_alloc(c, &c->d_ph, LRH_NSTAGE * c->ph_stride));
  if (cfg->second_fft_enable && c->timf2_mode == 1 && timf2_grid(cfg->fft1_n, cfg->max_batch) > 0) { c->ss_part_stride = (size_t)2 * timf2_grid(cfg->fft1_n, cfg->max_batch) * N1; A(dev_alloc(c, &c->d_ss_part, 2 * c->ss_part_stride)); }
  if (cfg->blanker_channels == 2) { A(dev_alloc(c, &c->d_pwr_sum, (size_t)cfg->timf2pow_size)); A(dev_alloc(c, &c->d_xbuf, (size_t)cfg->timf2pow_size)); A(dev_alloc(c, &c->d_xstat, 2));
    A(dev_alloc(c, &c->d_xbins, (size_t)2 * cfg->max_fft2n * N2)); A(dev_alloc(c, &c->d_xypower, (size_t)cfg->max_fft2n * N2));
    A(dev_alloc(c, &c->d_xysum, N2)); A(dev_alloc(c, &c->d_xysum_alt, N2)); }
  A(dev_alloc(c, &c->d_bln_tiles, (size_t)cfg->timf2pow_size / 16384 + 4)); A(dev_alloc(c, &c->d_bln_counts, (size_t)cfg->timf2pow_size / 16384 + 4));
  if (cfg->blanker_pulsewidth > 0) {                        // calibrated: guards chain the runs together, the long-run replay is the walk
    A(dev_alloc(c, &c->d_bln_wbusy, (size_t)cfg->timf2pow_size / 64 + 64)); A(dev_alloc(c, &c->d_bln_wstate, 2 * ((size_t)cfg->timf2pow_size / LRH_BLN_WTILE + 2)));
  }
  A(dev_alloc(c, &c->d_bst, 1)); A(dev_alloc(c, &c->d_partials, 2 * ((size_t)cfg->timf2pow_size / 1024 + cfg->timf2pow_size / 8192 + LRH_BLN_PARTIALS + 16)));
  if (rc == LRH_OK && lrh_host_malloc(&c->h_stage, LRH_STAGE_BYTES) != hipSuccess) { c->h_stage = nullptr; rc = fail(c, LRH_ENOMEM, "hipHostMalloc(staging)"); }
  if (rc == LRH_OK && lrh_host_malloc((void **)&c->h_ph, LRH_NSTAGE * c->ph_stride * sizeof(float)) != hipSuccess) rc = fail(c, LRH_ENOMEM, "hipHostMalloc");
  if (rc == LRH_OK && hipHostGetDevicePointer(&c->h_ph_dev, c->h_ph, 0) != hipSuccess) c->h_ph_dev = nullptr;
  for (int i = 0; i < LRH_NSTAGE && rc == LRH_OK; i++) if (hipEventCreateWithFlags(&c->ph_ev[i], hipEventDisableTiming) != hipSuccess) rc = LRH_EDEVICE;
  if (rc == LRH_OK) {
    hipStreamSynchronize(c->stream); c->opening = false;
    A(upload(c, c->d_window1, win1.data(), win1.size())); A(upload(c, c->d_invwin1, inv1.data(), N1));
    if (real1) {                               // k_fft1 stores the bare transform, k_realsplit applies the filter correction
      std::vector<float2> one(N1, make_float2(1.f, 0.f));
      A(dev_alloc(c, &c->d_unitcorr, N1, false)); A(upload(c, c->d_unitcorr, one.data(), N1));
    }
    A(upload(c, c->d_window2, c->h_window2.data(), N2)); A(upload(c, c->d_fqwin, c->h_fqwin.data(), c->Nm / 2 + 1));
    A(upload(c, c->d_mixwin, c->h_mixwin.data(), c->Nm / 2 + 1)); A(upload(c, c->d_sin2win, c->h_sin2win.data(), c->Nm)); A(upload(c, c->d_cos2win, c->h_cos2win.data(), c->Nm));
    if (cfg->fft3_n > 0) { A(upload(c, c->d_mix2win, c->h_mix2win.data(), c->Nm2 / 2 + 1)); A(upload(c, c->d_sin2win2, c->h_sin2win2.data(), c->Nm2)); A(upload(c, c->d_cos2win2, c->h_cos2win2.data(), c->Nm2)); }
    A(upload(c, c->d_yfac, c->h_yfac.data(), N1)); A(upload_filtercorr(c));
    A(upload(c, c->d_tw1, tw1.data(), N1)); A(upload(c, c->d_tw2, tw2.data(), N2)); A(upload(c, c->d_twm, twm.data(), c->Nm));
    if (cfg->fft2_n > 14) { A(upload(c, c->d_tw2a, tw2a.data(), tw2a.size())); A(upload(c, c->d_tw2b, tw2b.data(), tw2b.size())); }
    if (c->fft1_big) { A(upload(c, c->d_tw1a, tw1a.data(), tw1a.size())); A(upload(c, c->d_tw1b, tw1b.data(), tw1b.size())); }
    if (c->N3) {
      std::vector<float> ones(c->N3, 1.0f);
      A(upload(c, c->d_window3, win3.data(), c->N3)); A(upload(c, c->d_bgfilt, ones.data(), c->N3));
      A(upload(c, c->d_tw3, tw3.data(), c->N3)); A(upload(c, c->d_twm2, twm2.data(), c->Nm2));
    }
    A(upload(c, c->d_wf_itab, itab.data(), itab.size()));
    BlankState bs; memset(&bs, 0, sizeof bs);                          // buf.c:418-431, hires_graph.c:1157-1162
    bs.noise_floor = cfg->timf2_noise_floor; bs.amp_factor = 1.f; bs.despiked_pwr[0] = (float)cfg->timf2_noise_floor; bs.despiked_pwrinc[0] = 1;
    bs.limit = (unsigned int)((float)cfg->timf2_noise_floor * cfg->stupid_bln_factor);
    A(upload(c, c->d_bst, &bs, 1));
  }
#undef A
  if (rc != LRH_OK) { lrh_close(c); return rc; }
  c->ms.mix1_selfreq = -1; c->ms.mix1_point = -1; c->ms.mix1_old_point = 0;
  c->ms.mix1_phase = c->ms.mix1_phase_step = c->ms.mix1_phase_rot = c->ms.mix1_old_phase = 0;
  // all-weak routing until the control plane supplies liminfo
  std::vector<float> lim(N1, 0.f);
  *out = c;
  rc = lrh_set_liminfo(c, lim.data());
  c->pack_prev_stale = false;               // no transform before the first one
  c->have_liminfo = false;
  if (rc != LRH_OK) { *out = nullptr; lrh_close(c); }
  return rc;
}
LRH_CATCH_OPEN(out)

int lrh_get_derived(const lrh_ctx *c, int *i1, int *i2, int *ms, int *mi, int *t3b)
try {
  if (!c) return LRH_EINVAL;
  if (i1) *i1 = c->I1; if (i2) *i2 = c->I2; if (ms) *ms = c->Nm; if (mi) *mi = c->Im; if (t3b) *t3b = 2 * c->Mm;
  return LRH_OK;
}
LRH_CATCH(c)

void lrh_ptrs_init(const lrh_ctx *c, lrh_ptrs *p)
{
  (void)c; memset(p, 0, sizeof *p);
  p->fft1_lowlevel_fraction = .75f;        // buf.c:343
}

// device copy of the filter correction = table x channel phasing constant (both multiply every bin, one after the other,
// fft1.c:4064-4080 then 4119-4127)
static int upload_filtercorr(lrh_ctx *c)
{
  std::vector<float> eff(c->h_filtercorr);
  if (c->ch2_c1 != 1.0f || c->ch2_c2 != 0.0f)
    for (int i = 0; i < c->N1; i++) {
      const float a = c->h_filtercorr[2 * i], b = c->h_filtercorr[2 * i + 1];
      eff[2 * i] = a * c->ch2_c1 + b * c->ch2_c2;          // (a + jb)(c1 - j c2)
      eff[2 * i + 1] = b * c->ch2_c1 - a * c->ch2_c2;
    }
  HIPCHK(c, stage_h2d(c, c->d_filtercorr, eff.data(), 8 * c->N1, c->stream));
  std::vector<float> perm;
  if (c->d_filtercorr_v) {
    // k_fft1v: thread t ends its forward transform on the bins kk(t) + T j, j = 0..31 (T = N1 / 32; kk: Fft1vGeom / kk_of in lrh_kernels.hip);
    // the table in that order, [j][t], makes the 32 loads of a thread whole cache lines per wave
    const int T = c->N1 / 32, B2 = c->cfg.fft1_n - 10, R2 = 1 << B2, KB = 5 - B2, NKB = 1 << KB;
    perm.resize(2 * (size_t)c->N1);
    for (int j = 0; j < 32; j++)
      for (int t = 0; t < T; t++) {                        // [j / 2][t][j & 1]: two bins of a thread per 16-byte load, a wave's loads whole lines
        const int lam = t & 31, kk = (t >> 5) + R2 * (lam & (NKB - 1)) + 32 * (lam >> KB), f = kk + T * j;
        const size_t at = 2 * ((size_t)(j >> 1) * T + t) + (j & 1);
        perm[2 * at] = eff[2 * f]; perm[2 * at + 1] = eff[2 * f + 1];
      }
    HIPCHK(c, stage_h2d(c, c->d_filtercorr_v, perm.data(), 8 * c->N1, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}

int lrh_set_filtercorr(lrh_ctx *c, const float *fc)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  { const int rcj_ = join_handles(c); if (rcj_) return rcj_; }    // blocks the fft1_b workers have noted are transformed with the table they were handed over under
  if (fc) c->h_filtercorr.assign(fc, fc + 2 * c->N1); else default_filtercorr(c);
  c->f1_end_valid = false;
  return upload_filtercorr(c);
}
LRH_CATCH(c)

int lrh_set_ch2_phasing(lrh_ctx *c, float c1, float c2)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  { const int rcj_ = join_handles(c); if (rcj_) return rcj_; }    // (as in lrh_set_filtercorr: noted blocks go out under the old phasing)
  c->ch2_c1 = c1; c->ch2_c2 = c2;
  c->f1_end_valid = false;                               // the table the fused kernel's partner would be recomputed with has changed
  return upload_filtercorr(c);
}
LRH_CATCH(c)

int lrh_set_liminfo(lrh_ctx *c, const float *liminfo)
try {
  LRH_ENTER(c);
  if (!c || !liminfo) return LRH_EINVAL;
  // pack the weak flags per first-pass butterfly of the N1 transform (see k_timf2)
  const int R0 = c->cfg.fft1_n >= 10 ? 16 : 4;        // first-pass radix of the N1 transform (lrh_fft.hip.h)
  const int nb = c->fft1_big ? 0 : c->N1 / R0;
  std::vector<unsigned int> pack(c->N1, 0u);
  int low = 0;
  if (c->fft1_big)                                     // four-step timf2: dense bits, bit (k & 31) of word k >> 5
    for (int k = 0; k < c->N1; k++) if (liminfo[k] == 0) pack[k >> 5] |= 1u << (k & 31);
  for (int i = 0; i < nb; i++) {
    unsigned int m = 0;
    for (int s = 0; s < R0; s++) if (liminfo[i + s * nb] == 0) m |= 1u << s;
    pack[i] = m;
  }
  for (int i = 0; i < c->N1; i++) if (liminfo[i] == 0) low++;
  // d_pack_prev (routing of the transform before the next batch) is rolled forward by lrh_make_timf2
  HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2)); if (c->stream_sel) HIPCHK(c, hipStreamSynchronize(c->stream_sel));
  c->sel_table_pending = false;
  c->h_pack = pack;
  pack_new_table(c);
  HIPCHK(c, stage_h2d(c, c->d_pack_cur, c->h_pack.data(), 4 * c->N1, c->stream));
  HIPCHK(c, stage_h2d(c, c->d_liminfo, liminfo, 4 * c->N1, c->stream));   // the table lrh_fft1_update_liminfo carries on from
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->sel_pending = false;
  c->lowlevel_points = low;
  c->have_liminfo = true;
  return LRH_OK;
}
LRH_CATCH(c)

// fft1_update_liminfo + selfreq_liminfo on the device (include/linrad_hip.h); k_sellim / k_sellim2 (the routing words are their last step)
// installs the weak-bin count of update number `seq` (1-based) once its readback has arrived
static int sellim_install(lrh_ctx *c, unsigned seq)
{
  if (!c->sel_pending || seq == 0 || seq > c->sel_seq || seq + 3 <= c->sel_seq) return LRH_OK;
  HIPCHK(c, hipEventSynchronize(c->ev_sel_slot[seq % 3]));
  c->lowlevel_points = c->h_sel_low[seq % 3];
  if (seq == c->sel_seq) c->sel_pending = false;
  return LRH_OK;
}
// the same without ever waiting: the newest update whose readback has arrived (the host runs rounds ahead of the device; a wait
// here would let the device run dry once per round)
static void sellim_poll(lrh_ctx *c)
{
  if (!c->sel_pending) return;
  for (unsigned back = 0; back < 3 && back < c->sel_seq; back++) {
    const unsigned seq = c->sel_seq - back;
    if (hipEventQuery(c->ev_sel_slot[seq % 3]) == hipSuccess) {
      c->lowlevel_points = c->h_sel_low[seq % 3];
      if (back == 0) c->sel_pending = false;
      return;
    }
  }
}
// what both limiter kernels need from the parameter block and the context
static int sellim_args(lrh_ctx *c, const lrh_sellim *q, SellimArgs *out)
{
  SellimArgs a; memset(&a, 0, sizeof a);
  a.slowsum = c->d_slowsum; a.yfac = c->d_yfac; a.liminfo = c->d_liminfo; a.old_liminfo = c->d_old_liminfo; a.tmp = c->d_sel_tmp;
  a.wait = c->d_sel_wait; a.pack = c->d_pack_cur; a.st = c->d_sel_st;
  a.n = c->N1; a.n2 = c->N2; a.avg1 = c->cfg.fft_avg1num; a.r0 = c->fft1_big ? 0 : (c->cfg.fft1_n >= 10 ? 16 : 4);
  a.maxlevel = q->sellim_maxlevel; a.spek_avgnum = q->spek_avgnum; a.blocktime = q->fft1_blocktime; a.ston = q->blanker_ston_fft1;
  a.par2 = q->sellim_par2; a.par3 = q->sellim_par3; a.par4 = q->sellim_par4; a.par5 = q->sellim_par5; a.par6 = q->sellim_par6;
  a.par7 = q->sellim_par7; a.par8 = q->sellim_par8; a.group_points = q->liminfo_group_points;
  a.first_point = q->fft1_first_point; a.last_point = q->fft1_last_point; a.first_inband = q->fft1_first_inband; a.last_inband = q->fft1_last_inband;
  a.bw_fftxpts = q->baseband_bw_fftxpts; a.ston_scale = q->ston_scale;
  a.selfreq = c->ms.mix1_selfreq; a.points_per_hz = c->cfg.fftx_points_per_hz; a.second_fft = c->cfg.second_fft_enable;
  a.bst = c->d_bst; a.desired = nullptr; a.desired_totsum = 0;
  if (c->fft1_big) {                                       // table and group minima in global memory (k_sellim<true>)
    if (!c->d_sel_bigb) { int rc = dev_alloc(c, &c->d_sel_bigb, c->N1); if (!rc) rc = dev_alloc(c, &c->d_sel_bigg, c->N1 / 4 + 8); if (rc) return rc; HIPCHK(c, hipStreamSynchronize(c->stream)); }
    a.big_b = c->d_sel_bigb; a.big_g = c->d_sel_bigg;
  }
  { static const int dbg = getenv("LRH_SELLIM_DEBUG") ? atoi(getenv("LRH_SELLIM_DEBUG")) : 0; a.debug = dbg; }
  if (q->fft1_desired) {                                  // calibrated amplitude factor: the table travels once (and again when it changes)
    if (c->h_sel_desired.size() != (size_t)c->N1 || memcmp(c->h_sel_desired.data(), q->fft1_desired, 4 * (size_t)c->N1)) {
      c->h_sel_desired.assign(q->fft1_desired, q->fft1_desired + c->N1);
      if (!c->d_sel_desired) { const int rc = dev_alloc(c, &c->d_sel_desired, c->N1); if (rc) return rc; }
      if (c->stream_sel) HIPCHK(c, hipStreamSynchronize(c->stream_sel));
      HIPCHK(c, stage_h2d(c, c->d_sel_desired, c->h_sel_desired.data(), 4 * (size_t)c->N1, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      float tot = 0;
      for (int i = 0; i < c->N1; i++) tot += q->fft1_desired[i] * q->fft1_desired[i];
      c->sel_desired_totsum = tot;
    }
    a.desired = c->d_sel_desired; a.desired_totsum = c->sel_desired_totsum;
  }
  a.powersum2 = c->d_powersum2; a.ston2 = q->blanker_ston_fft2; a.blocktime2 = q->fft2_blocktime; a.wf_avgnum = c->cfg.waterfall_avgnum;
  *out = a;
  return LRH_OK;
}
// `which` 1: fft1_update_liminfo, 2: fft2_update_liminfo
static int sellim_run(lrh_ctx *c, lrh_ptrs *p, const lrh_sellim *q, int which)
{
  if (!c->h_sel_low) {
    if (lrh_host_malloc((void **)&c->h_sel_low, 3 * sizeof(int)) != hipSuccess) return fail(c, LRH_ENOMEM, "hipHostMalloc");
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_sel, hipEventDisableTiming));
    for (hipEvent_t &e : c->ev_sel_slot) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  sellim_poll(c);                                           // the count of the newest finished update (unless exact_stats: see the end)
  SellimArgs a;
  { const int rc = sellim_args(c, q, &a); if (rc) return rc; }
  if (!c->stream_sel) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream_sel, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_sel_wait2, hipEventDisableTiming));
  }
  hipStream_t S = c->stream_sel;
  if (which == 1 && c->clv_defer) {
    // deferred linear blanker: the search of this round is issued a round from now and must see the amplitude factor as it is BEFORE this
    // run (the serial order) -- a copy, taken on this stream ahead of the run, in the slot that search will read
    if (!c->d_clv_amp) { const int rc = dev_alloc(c, &c->d_clv_amp, 2); if (rc) return rc; HIPCHK(c, hipEventCreateWithFlags(&c->ev_amp, hipEventDisableTiming)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    HIPCHK(c, hipMemcpyAsync(c->d_clv_amp + (c->clv_amp_seq & 1), (char *)c->d_bst + offsetof(BlankState, amp_factor), sizeof(float), hipMemcpyDeviceToDevice, S));
    HIPCHK(c, hipEventRecord(c->ev_amp, S));
    c->clv_amp_seq++;
  }
  // the block at the advanced pointer (sellim.c:788, fft1.c:4519): the reference looks right after fft1_c has closed a period, when that slot
  // still holds the sums of a ring lap ago.  A look in the middle of a period (batched rounds whose length is no multiple of
  // fft_avg1num) would find the unfinished sums of the period in progress there, a spectrum at a fraction of its level: it takes
  // the newest finished period instead.
  if (which == 1) a.sumsq = c->d_sumsq + ((p->fft1_sumsq_pa - (p->fft1_sumsq_counter ? c->N1 : 0) + c->cfg.fft1_sumsq_bufsize) & c->sumsq_mask);
  else {
    if (!c->d_sel_ftmp) { const int rc = dev_alloc(c, &c->d_sel_ftmp, c->N1); if (rc) return rc; HIPCHK(c, hipStreamSynchronize(c->stream)); }
    a.tmp = c->d_sel_ftmp;
    a.par1 = q->sellim_par1;
    if (a.par1 == 1) {                                      // the region list of variant 1 (reg_noise, reg_first_point, reg_length: buf.c:976-980), zero at first
      const size_t cap = (size_t)c->N1 / q->liminfo_group_points + 8;
      if (c->sel_reg_cap < cap) {
        if (c->stream_sel) HIPCHK(c, hipStreamSynchronize(c->stream_sel));
        if (c->d_sel_reg) { (void)lrh_dev_free(c->d_sel_reg); c->d_sel_reg = nullptr; c->sel_reg_cap = 0; }
        const int rc = dev_alloc(c, &c->d_sel_reg, 3 * cap); if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->sel_reg_cap = cap;
      }
      a.reg_noise = c->d_sel_reg; a.reg_first = (int *)(c->d_sel_reg + c->sel_reg_cap); a.reg_len = (int *)(c->d_sel_reg + 2 * c->sel_reg_cap);
    }
  }
  // behind everything queued so far on the main stream (the previous make_timf2 that read the routing words; the sums in the serial
  // order) and, when this round's sums (first limiter) or the fft2 power sums (second) ran there, on the side stream -- not otherwise:
  // the blanker of the previous round is queued there too and would hold the table back for nothing
  if (c->last_main_ev) HIPCHK(c, hipStreamWaitEvent(S, c->last_main_ev, 0));
  else { HIPCHK(c, hipEventRecord(c->ev_sel_wait, c->stream)); HIPCHK(c, hipStreamWaitEvent(S, c->ev_sel_wait, 0)); }
  if (which == 2 || c->sums_stream == c->stream2) { HIPCHK(c, hipEventRecord(c->ev_sel_wait2, c->stream2)); HIPCHK(c, hipStreamWaitEvent(S, c->ev_sel_wait2, 0)); }
  pack_new_table(c); a.pack = c->d_pack_cur;                 // the new table's routing words go beside the ones the last make_timf2 used
  { hipStream_t keep = c->cur; c->cur = S;
    { ProfScope ps(c, "sellim"); hipError_t e_ = which == 1 ? launch_sellim(a, S) : launch_sellim2(a, S);
      if (e_ != hipSuccess) { c->cur = keep; return fail(c, LRH_EDEVICE, "launch_sellim", e_); } }
    c->cur = keep; }
  c->sel_seq++;
  HIPCHK(c, hipEventRecord(c->ev_sel, S));                   // the table is there: the next make_timf2 does not wait for the read-back below
  HIPCHK(c, hipMemcpyAsync(&c->h_sel_low[c->sel_seq % 3], &c->d_sel_st->low, sizeof(int), hipMemcpyDeviceToHost, S));
  HIPCHK(c, hipEventRecord(c->ev_sel_slot[c->sel_seq % 3], S));
  c->sel_pending = true; c->sel_table_pending = true; c->have_liminfo = true;
  if (q->exact_stats) return sellim_install(c, c->sel_seq);
  return LRH_OK;
}
// One gate for every way the limiter's parameters reach the kernels (both update entry points, lrh_wideband_limiter): the kernels
// index LDS and global arrays with these values and divide by group_points, so nothing unchecked may pass.  `second`: the
// fft2 variant's extra demands (groups of at least 16 bins, a block time for the hold-off).
static int sellim_check(lrh_ctx *c, const lrh_sellim *q, bool second)
{
  if (q->struct_size != (int)sizeof *q) return LRH_EINVAL;
  if (c->N1 > 32768) return fail(c, LRH_EINVAL, "selective limiter on the device: fft1_size <= 32768; use lrh_set_liminfo");
  const int gp = q->liminfo_group_points;
  const bool ok = gp >= (second ? 16 : 4) && gp <= c->N1 &&
                  q->fft1_first_point >= 0 && q->fft1_first_point <= q->fft1_last_point && q->fft1_last_point < c->N1 &&
                  q->fft1_first_inband >= 0 && q->fft1_first_inband <= q->fft1_last_inband && q->fft1_last_inband < c->N1 &&
                  q->sellim_maxlevel >= 1 && q->spek_avgnum >= 1 && q->fft1_blocktime > 0 && (!second || q->fft2_blocktime > 0);
  if (!ok) return fail(c, LRH_EINVAL, "selective limiter: parameter out of range (group_points, first/last point or inband, maxlevel, spek_avgnum, blocktime)");
  if (second && (!c->cfg.second_fft_enable || c->cfg.blanker_channels == 2 || c->N2 < c->N1))
    return fail(c, LRH_EINVAL, "fft2_update_liminfo: one channel, second fft on, fft2_size >= fft1_size");
  if (second && (q->sellim_par1 < 0 || q->sellim_par1 > 2)) return fail(c, LRH_EINVAL, "fft2_update_liminfo: sellim_par1 is 0, 1 or 2 (hires_graph.c:1175)");
  return LRH_OK;
}
int lrh_fft1_update_liminfo(lrh_ctx *c, lrh_ptrs *p, const lrh_sellim *q)
try {
  LRH_ENTER(c);
  if (!c || !p || !q) return LRH_EINVAL;
  if (c->rec) return fail(c, LRH_ESTATE, "not inside lrh_wideband_dsp");
  { const int rc = sellim_check(c, q, false); if (rc) return rc; }
  return sellim_run(c, p, q, 1);
}
LRH_CATCH(c)
int lrh_fft2_update_liminfo(lrh_ctx *c, lrh_ptrs *p, const lrh_sellim *q)
try {
  LRH_ENTER(c);
  if (!c || !p || !q) return LRH_EINVAL;
  if (c->rec) return fail(c, LRH_ESTATE, "not inside lrh_wideband_dsp");
  { const int rc = sellim_check(c, q, true); if (rc) return rc; }
  return sellim_run(c, p, q, 2);
}
LRH_CATCH(c)
int lrh_wideband_limiter(lrh_ctx *c, const lrh_sellim *par, int fft2_too)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (c->rec) return fail(c, LRH_ESTATE, "not inside lrh_wideband_dsp");
  c->wl_on = false;
  if (!par) return LRH_OK;
  // a coupled pair's round-level order (dsp_coupled) makes no limiter calls: the reference's two-channel limiter works on the channels' summed
  // spectra with its limit scaled by rx_rf_channels, which is not built -- said here, not ignored there
  if (c->cfg.blanker_channels == 2) return fail(c, LRH_ESTATE, "lrh_wideband_limiter: not with two coupled channels (the limiter calls are the caller's there)");
  { int rc = sellim_check(c, par, false); if (!rc && fft2_too) rc = sellim_check(c, par, true); if (rc) return rc; }
  c->wl_par = *par;
  if (par->fft1_desired) { c->wl_desired.assign(par->fft1_desired, par->fft1_desired + c->N1); c->wl_par.fft1_desired = c->wl_desired.data(); }
  c->wl_on = true; c->wl_fft2 = fft2_too != 0; c->wl_cnt1 = 0; c->wl_cnt2 = 0;
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_get_liminfo_amplitude_factor(lrh_ctx *c, float *f)
try {
  LRH_ENTER(c);
  if (!c || !f) return LRH_EINVAL;
  if (c->sel_table_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_sel, 0));
  HIPCHK(c, stage_d2h(c, f, (char *)c->d_bst + offsetof(BlankState, amp_factor), sizeof(float), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_set_liminfo_amplitude_factor(lrh_ctx *c, float f)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (c->rec) return fail(c, LRH_ESTATE, "not inside lrh_wideband_dsp");
  HIPCHK(c, hipStreamSynchronize(c->stream2)); if (c->stream_sel) HIPCHK(c, hipStreamSynchronize(c->stream_sel));
  HIPCHK(c, stage_h2d(c, (char *)c->d_bst + offsetof(BlankState, amp_factor), &f, sizeof(float), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_get_liminfo(lrh_ctx *c, float *dst)
try {
  LRH_ENTER(c);
  if (!c || !dst) return LRH_EINVAL;
  if (c->sel_table_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_sel, 0));
  HIPCHK(c, stage_d2h(c, dst, c->d_liminfo, 4 * c->N1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)

// ---- linear ("clever") blanker tables (include/linrad_hip.h; init_blanker's products, buf.c:1786-2057)
int lrh_set_blanker_tables(lrh_ctx *c, const lrh_blanker_tables *t)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (c->rec) return fail(c, LRH_ESTATE, "not inside lrh_wideband_dsp");
  HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2)); if (c->stream_sel) HIPCHK(c, hipStreamSynchronize(c->stream_sel));
  for (void **q_ : { (void **)&c->d_bt_refpulse, (void **)&c->d_bt_phasefunc, (void **)&c->d_bt_pulindex, (void **)&c->d_bln_flag, (void **)&c->d_bln_cand, (void **)&c->d_xweak, (void **)&c->d_tf_partner })
    if (*q_) { lrh_dev_free(*q_); *q_ = nullptr; }
  c->clever_on = false;
  if (!t) return LRH_OK;
  const int rs = t->refpul_size, pw = c->cfg.blanker_pulsewidth;
  if (t->clever_bln_mode < 1 || t->clever_bln_mode > 2 || rs < 4 || rs > 256 || (rs & (rs - 1)) || t->largest_blnfit < 0 ||
      t->largest_blnfit >= LRH_BLN_INFO_SIZE || !t->refpulse || !t->phasefunc || !t->pulindex || pw < 1 || 2 * pw >= rs)
    return fail(c, LRH_EINVAL, "linear blanker: bad table sizes");
  for (int i = 0; i <= t->largest_blnfit; i++)
    if (t->bln[i].size < 4 || t->bln[i].size > rs || (t->bln[i].size & 1) || (i && t->bln[i].size <= t->bln[i - 1].size)) return fail(c, LRH_EINVAL, "linear blanker: bad bln[] sizes");
  if (c->cfg.blnfit_range != t->bln[t->largest_blnfit].size / 2 + pw) return fail(c, LRH_EINVAL, "cfg.blnfit_range != bln[largest_blnfit].size/2 + blanker_pulsewidth (buf.c:2057)");
  for (int i = 0; i < LRH_MAX_REFPULSES; i++) if (t->pulindex[i] < 0 || t->pulindex[i] >= LRH_MAX_REFPULSES) return fail(c, LRH_EINVAL, "linear blanker: pulindex out of range");
  if (c->cfg.timf2pow_size < 1024) return fail(c, LRH_EINVAL, "linear blanker: timf2pow_size < 1024");
  const size_t nr = (size_t)2 * LRH_MAX_REFPULSES * rs;
  int rc = LRH_OK;
  if ((rc = dev_alloc(c, &c->d_bt_refpulse, nr)) || (rc = dev_alloc(c, &c->d_bt_phasefunc, (size_t)2 * rs)) || (rc = dev_alloc(c, &c->d_bt_pulindex, LRH_MAX_REFPULSES)) ||
      (rc = dev_alloc(c, &c->d_bln_flag, (size_t)c->cfg.timf2pow_size)) || (rc = dev_alloc(c, &c->d_bln_cand, (size_t)c->cfg.timf2pow_size / 64))) return rc;
  if (c->cfg.blanker_channels == 2 && ((rc = dev_alloc(c, &c->d_xweak, (size_t)2 * c->cfg.timf2pow_size)) || (rc = dev_alloc(c, &c->d_tf_partner, (size_t)c->cfg.timf2pow_size)))) return rc;
  HIPCHK(c, stage_h2d(c, c->d_bt_refpulse, t->refpulse, 4 * nr, c->stream));
  HIPCHK(c, stage_h2d(c, c->d_bt_phasefunc, t->phasefunc, 8 * (size_t)rs, c->stream));
  HIPCHK(c, stage_h2d(c, c->d_bt_pulindex, t->pulindex, 4 * LRH_MAX_REFPULSES, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_bln_flag, 0, c->cfg.timf2pow_size, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_bln_cand, 0, c->cfg.timf2pow_size / 8, c->stream));
  HIPCHK(c, stage_h2d(c, (char *)c->d_bst + offsetof(BlankState, clever_limit), &t->clever_bln_limit, sizeof(unsigned int), c->stream));
  HIPCHK(c, stage_h2d(c, (char *)c->d_bst + offsetof(BlankState, amp_factor), &t->liminfo_amplitude_factor, sizeof(float), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->bt = *t; c->bt.refpulse = nullptr; c->bt.phasefunc = nullptr; c->bt.pulindex = nullptr;
  c->clever_on = true;
  return LRH_OK;
}
LRH_CATCH(c)

// ---- spur subtraction (include/linrad_hip.h): configuration and the control plane's hand-over
int lrh_spur_config(lrh_ctx *c, int max_spurs, int speknum, const float *spectra)
try {
  LRH_ENTER(c);
  // the ring the spurs live in (fftx of spur.c): the fft2 transforms, or -- second fft off, fft1_c's AFC branch (fft1.c:4196-4244) -- the fft1 transforms
  const bool second = c && c->cfg.second_fft_enable != 0;
  if (!c || max_spurs < 0 || (max_spurs && (!spectra || speknum < 4 || 4 * speknum > (second ? c->cfg.max_fft2n : c->cfg.max_fft1n) || speknum > 1022))) return LRH_EINVAL;   // 1022: k_spur keeps speknum + 2 history entries in LDS (107 KB then)
  if (c->cfg.blanker_channels == 2) return fail(c, LRH_ESTATE, "spur subtraction: one channel");
  if (!second && c->cfg.fft1_float_sparse) return fail(c, LRH_ESTATE, "spur subtraction on the fft1 transforms: whole spectra in the ring");
  c->spur_ring = second ? c->d_fft2 : c->d_fft1; c->spur_nx = second ? c->N2 : c->N1; c->spur_maxn = second ? c->cfg.max_fft2n : c->cfg.max_fft1n;
  c->spur_ff = second ? (float)c->M2 / (float)c->N2 : (float)c->M1 / (float)c->N1;
  HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2)); if (c->stream_sel) HIPCHK(c, hipStreamSynchronize(c->stream_sel));
  for (void **q_ : { (void **)&c->d_spurs, (void **)&c->d_spur_table, (void **)&c->d_spur_signal, (void **)&c->d_spur_touched, (void **)&c->d_spur_spectra, (void **)&c->d_spur_ind })
    if (*q_) { lrh_dev_free(*q_); *q_ = nullptr; }
  c->spur_max = 0; c->spur_n = 0; c->spur_speknum = 0;
  if (!max_spurs) return LRH_OK;
  const size_t maxn = c->spur_maxn;
  int rc = LRH_OK;
  if ((rc = dev_alloc(c, &c->d_spurs, max_spurs)) || (rc = dev_alloc(c, &c->d_spur_table, max_spurs * maxn * 14)) || (rc = dev_alloc(c, &c->d_spur_signal, max_spurs * maxn * 2)) ||
      (rc = dev_alloc(c, &c->d_spur_ind, max_spurs * maxn)) || (rc = dev_alloc(c, &c->d_spur_touched, 2 * max_spurs + 2)) || (rc = dev_alloc(c, &c->d_spur_spectra, LRH_SPUR_SPECTRA))) return rc;
  HIPCHK(c, stage_h2d(c, c->d_spur_spectra, spectra, sizeof(float) * LRH_SPUR_SPECTRA, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->spur_max = max_spurs; c->spur_speknum = speknum;
  return LRH_OK;
}
LRH_CATCH(c)
// The search for new spurs on the resident power rows (include/linrad_hip.h): make_fft2 keeps spursearch_powersum over 3 spur_speknum
// transforms and, with the next one, forms spursearch_spectrum and cleans it (fft2.c:673-699, spursub.c:40-175).  The cleanup runs on a
// stream of its own behind the transform that completed the sums: it is one workgroup's serial walk, not main-stream work.
int lrh_spur_search_config(lrh_ctx *c, int first_point, int last_point)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (!c->spur_max) return fail(c, LRH_ESTATE, "lrh_spur_config first");
  if (c->stream_ss) HIPCHK(c, hipStreamSynchronize(c->stream_ss));
  for (float **q_ : { &c->d_ss_sum, &c->d_ss_spec_base, &c->d_ss_min, &c->d_ss_out }) if (*q_) { lrh_dev_free(*q_); *q_ = nullptr; }
  c->ss_counter = 0; c->ss_completed = 0; c->ss_busy = false;
  if (first_point == 0 && last_point == 0) return LRH_OK;
  if (first_point < 0 || last_point >= c->spur_nx || last_point - first_point < 64) return LRH_EINVAL;
  if (c->cfg.second_fft_enable && c->cfg.fft2_float_sparse) return fail(c, LRH_ESTATE, "the spur search reads whole power rows: cfg.fft2_float_sparse must be 0");
  int rc;
  // (the reference's walk reads up to three bins before the first and 31 behind the last point of the range, spursub.c:48, 160-167)
  if ((rc = dev_alloc(c, &c->d_ss_sum, (size_t)c->spur_nx + 64)) || (rc = dev_alloc(c, &c->d_ss_spec_base, (size_t)c->spur_nx + 128)) ||
      (rc = dev_alloc(c, &c->d_ss_min, (size_t)c->spur_nx / 32 + 64)) || (rc = dev_alloc(c, &c->d_ss_out, 4))) return rc;
  if (!c->stream_ss) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream_ss, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_ss_in, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_ss_done, hipEventDisableTiming));
  }
  c->ss_first = first_point; c->ss_last = last_point;
  return LRH_OK;
}
LRH_CATCH(c)
// one new power row (transform `na` of the ring), behind whatever `src` has enqueued so far
static int spur_search_row(lrh_ctx *c, int na, hipStream_t src)
{
  if (!c->d_ss_sum) return LRH_OK;
  SpurSearchArgs a; memset(&a, 0, sizeof a);
  a.sum = c->d_ss_sum; a.spec = c->d_ss_spec_base + 32; a.mins = c->d_ss_min; a.z = c->spur_ring + (size_t)na * c->spur_nx;
  a.first = c->ss_first; a.last = c->ss_last; a.spectra = c->d_spur_spectra; a.out = c->d_ss_out;
  const double s3 = sqrt((double)(float)(3 * c->spur_speknum));
  a.noise_factor = pow(10., 0.7 / s3); a.thr_factor = pow(10., 1.5 / s3);
  // the rows are element-wise and short: on the caller's stream, right behind the kernels that made the power row.  Only the cleanup
  // goes to the stream of its own; the row that forms the next search spectrum waits for it (3 spur_speknum transforms later: long done)
  if (c->ss_counter > 3 * c->spur_speknum) {
    c->ss_counter = 0;
    a.mode = 2;
    if (c->ss_busy) { HIPCHK(c, hipStreamWaitEvent(src, c->ev_ss_done, 0)); c->ss_busy = false; }
    HIPCHK(c, launch_spur_search_row(a, src));
    HIPCHK(c, hipEventRecord(c->ev_ss_in, src)); HIPCHK(c, hipStreamWaitEvent(c->stream_ss, c->ev_ss_in, 0));
    HIPCHK(c, launch_spur_search_cleanup(a, c->stream_ss));
    HIPCHK(c, hipEventRecord(c->ev_ss_done, c->stream_ss)); c->ss_busy = true;
    c->ss_completed++;
  } else {
    a.mode = c->ss_counter == 0 ? 0 : 1;
    HIPCHK(c, launch_spur_search_row(a, src));
    c->ss_counter++;
  }
  return LRH_OK;
}
int lrh_spur_search_get(lrh_ctx *c, float *spectrum, float *threshold, int *completed, int *sum_counter)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (!c->d_ss_sum) return fail(c, LRH_ESTATE, "lrh_spur_search_config first");
  HIPCHK(c, hipStreamSynchronize(c->stream_ss));
  if (spectrum) HIPCHK(c, stage_d2h(c, spectrum, c->d_ss_spec_base + 32 + c->ss_first, sizeof(float) * (size_t)(c->ss_last - c->ss_first + 1), c->stream));
  if (threshold) { float o[2] = { 0, 0 }; HIPCHK(c, stage_d2h(c, o, c->d_ss_out, sizeof o, c->stream)); *threshold = o[0]; }
  if (completed) *completed = c->ss_completed;
  if (sum_counter) *sum_counter = c->ss_counter;
  return LRH_OK;
}
LRH_CATCH(c)

// what k_spur / k_spur_acquire need from the context: the loop constants follow spur_speknum (buf.c:480, 1141-1170)
static void spur_args(lrh_ctx *c, SpurArgs *out, int first_na, int batch)
{
  SpurArgs sa; memset(&sa, 0, sizeof sa);
  const int N = c->spur_nx;
  sa.fft2 = c->spur_ring; sa.n2 = N; sa.first_na = first_na; sa.na_mask = c->spur_maxn - 1; sa.batch = batch;
  sa.nspurs = c->spur_n; sa.speknum = c->spur_speknum; sa.numsub = sa.speknum - 1; sa.avgnum = sa.speknum / 3; if (sa.avgnum > 10) sa.avgnum = 10;
  sa.freq_factor = c->spur_ff;
  sa.max_d2 = (float)(PI_L * sa.freq_factor / sa.speknum);
  sa.minston = (float)(1 / sqrt(0.5 * (float)(sa.speknum)));
  { float t1 = (float)(0.5 * sa.speknum); sa.weiold = t1 / (1 + t1); sa.weinew = 1 / (1 + t1);
    t1 = (float)(-0.5 * sa.numsub); sa.linefit = 0; for (int i = 0; i < sa.speknum; i++) { sa.linefit += t1 * t1; t1 += 1; } }
  sa.spectra = c->d_spur_spectra; sa.spurs = c->d_spurs; sa.table = c->d_spur_table; sa.signal = c->d_spur_signal; sa.ind = c->d_spur_ind; sa.touched = c->d_spur_touched;
  *out = sa;
}
// store_new_spur + spur_phase_lock (spursub.c:619, 1247) on the resident spectra: the control plane names the seven bins, the device
// takes the history, reads the frequency off it and closes the loop; one int comes back
int lrh_spur_acquire(lrh_ctx *c, const lrh_ptrs *p, int pnt, int *locked)
try {
  LRH_ENTER(c);
  if (!c || !p || !locked) return LRH_EINVAL;
  LRH_WRITES(c, RB(LRH_RING_FFT2_FLOAT) | RB(LRH_RING_FFT1_FLOAT));
  *locked = 0;
  if (!c->spur_max) return fail(c, LRH_ESTATE, "lrh_spur_config first");
  if (c->spur_n >= c->spur_max || pnt < 1 || pnt + 9 > c->spur_nx) return LRH_EINVAL;
  if (c->cfg.second_fft_enable && c->cfg.fft2_float_sparse) return fail(c, LRH_ESTATE, "spur acquisition reads whole fft2 transforms: cfg.fft2_float_sparse must be 0");
  if (!c->cfg.second_fft_enable) { const int rcj_ = join_handles(c); if (rcj_) return rcj_; }      // (the fft1_b workers' transforms)
  SpurArgs sa; spur_args(c, &sa, p->fft2_na, 0);
  HIPCHK(c, launch_spur_acquire(sa, pnt, c->d_spur_touched + 2 * c->spur_max, c->stream));
  int res = 0;
  HIPCHK(c, stage_d2h(c, &res, c->d_spur_touched + 2 * c->spur_max, sizeof res, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (res) { c->spur_n++; *locked = 1; }
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_spur_set(lrh_ctx *c, int n, const lrh_spur *sp, const float *table, const float *signal, const int *ind)
try {
  LRH_ENTER(c);
  if (!c || n < 0 || n > c->spur_max || (n && (!sp || !table || !signal || !ind))) return LRH_EINVAL;
  if (c->rec) return fail(c, LRH_ESTATE, "not inside lrh_wideband_dsp");
  const size_t maxn = c->spur_maxn;
  if (n) {
    HIPCHK(c, stage_h2d(c, c->d_spurs, sp, n * sizeof *sp, c->stream));
    HIPCHK(c, stage_h2d(c, c->d_spur_table, table, n * maxn * 14 * sizeof(float), c->stream));
    HIPCHK(c, stage_h2d(c, c->d_spur_signal, signal, n * maxn * 2 * sizeof(float), c->stream));
    HIPCHK(c, stage_h2d(c, c->d_spur_ind, ind, n * maxn * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->spur_n = n;
  return LRH_OK;
}
LRH_CATCH(c)
// remove_spur / swap_spurs (spur.c:596-631, spursub.c:755): the control plane drops a spur or reorders the list; loop state and histories follow on the device
int lrh_spur_permute(lrh_ctx *c, int n, const int *src)
try {
  LRH_ENTER(c);
  if (!c || n < 0 || n > c->spur_n || (n && !src)) return LRH_EINVAL;
  if (c->rec) return fail(c, LRH_ESTATE, "not inside lrh_wideband_dsp");
  for (int i = 0; i < n; i++) if (src[i] < 0 || src[i] >= c->spur_n) return LRH_EINVAL;
  const size_t maxn = c->spur_maxn;
  const int old_n = c->spur_n;
  if (n) {
    std::vector<lrh_spur> sp(old_n); std::vector<float> tab((size_t)old_n * maxn * 14), sig((size_t)old_n * maxn * 2); std::vector<int> ind((size_t)old_n * maxn);
    HIPCHK(c, stage_d2h(c, sp.data(), c->d_spurs, old_n * sizeof(lrh_spur), c->stream));
    HIPCHK(c, stage_d2h(c, tab.data(), c->d_spur_table, tab.size() * sizeof(float), c->stream));
    HIPCHK(c, stage_d2h(c, sig.data(), c->d_spur_signal, sig.size() * sizeof(float), c->stream));
    HIPCHK(c, stage_d2h(c, ind.data(), c->d_spur_ind, ind.size() * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<lrh_spur> sp2(n); std::vector<float> tab2((size_t)n * maxn * 14), sig2((size_t)n * maxn * 2); std::vector<int> ind2((size_t)n * maxn);
    for (int i = 0; i < n; i++) {
      sp2[i] = sp[src[i]];
      memcpy(&tab2[(size_t)i * maxn * 14], &tab[(size_t)src[i] * maxn * 14], maxn * 14 * sizeof(float));
      memcpy(&sig2[(size_t)i * maxn * 2], &sig[(size_t)src[i] * maxn * 2], maxn * 2 * sizeof(float));
      memcpy(&ind2[(size_t)i * maxn], &ind[(size_t)src[i] * maxn], maxn * sizeof(int));
    }
    HIPCHK(c, stage_h2d(c, c->d_spurs, sp2.data(), n * sizeof(lrh_spur), c->stream));
    HIPCHK(c, stage_h2d(c, c->d_spur_table, tab2.data(), tab2.size() * sizeof(float), c->stream));
    HIPCHK(c, stage_h2d(c, c->d_spur_signal, sig2.data(), sig2.size() * sizeof(float), c->stream));
    HIPCHK(c, stage_h2d(c, c->d_spur_ind, ind2.data(), ind2.size() * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->spur_n = n;
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_spur_get(lrh_ctx *c, int max, lrh_spur *sp, int *n)
try {
  LRH_ENTER(c);
  if (!c || !sp || !n || max < 0) return LRH_EINVAL;
  *n = c->spur_n < max ? c->spur_n : max;
  if (*n) {
    HIPCHK(c, stage_d2h(c, sp, c->d_spurs, *n * sizeof *sp, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_set_waterfall_yfac(lrh_ctx *c, const float *y)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (y) c->h_yfac.assign(y, y + c->N1); else default_yfac(c);
  HIPCHK(c, stage_h2d(c, c->d_yfac, c->h_yfac.data(), 4 * c->N1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_get_table(lrh_ctx *c, const char *name, float *dst, int count)
try {
  LRH_ENTER(c);
  if (!c || !name || !dst) return LRH_EINVAL;
  const std::vector<float> *src = nullptr;
  if (!strcmp(name, "fft1_window")) src = &c->h_window1_ref;
  else if (!strcmp(name, "fft2_window")) src = &c->h_window2;
  else if (!strcmp(name, "mix1_fqwin")) src = &c->h_fqwin;
  else if (!strcmp(name, "fft1_filtercorr")) src = &c->h_filtercorr;
  else if (!strcmp(name, "wg_waterf_yfac")) src = &c->h_yfac;
  else if (!strcmp(name, "fft1_inverted_window")) src = &c->h_invwin1_ref;
  else if (!strcmp(name, "fft3_window")) src = &c->h_window3_ref;
  else return LRH_EINVAL;
  if (count > (int)src->size()) count = (int)src->size();
  memcpy(dst, src->data(), 4 * (size_t)count);
  return count;
}
LRH_CATCH(c)

static int flush_input_locked(lrh_ctx *c);
int lrh_timf1_write(lrh_ctx *c, const void *src, int off, int nbytes)
try {
  LRH_LOCK(c);
  if (!c || !src || nbytes < 0 || nbytes > c->cfg.timf1_bytes) return LRH_EINVAL;
  { std::lock_guard<std::mutex> lk_in(c->mtx_in); if (!c->in_spans.empty()) { const int rc_ = flush_input_locked(c); if (rc_) return rc_; } }   // (noted producer copies keep their place in front of this one)
  off &= c->timf1_bytemask;
  const char *s = (const char *)src; char *d = (char *)c->d_timf1;
  int first = nbytes < c->cfg.timf1_bytes - off ? nbytes : c->cfg.timf1_bytes - off;
  HIPCHK(c, stage_h2d(c, d + off, s, first, c->stream));
  if (nbytes > first) HIPCHK(c, stage_h2d(c, d, s + first, nbytes - first, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));      // the caller may reuse src
  return LRH_OK;
}
LRH_CATCH(c)
void *lrh_timf1_device_ptr(lrh_ctx *c) { return c ? c->d_timf1 : nullptr; }

// `bytes` more have been issued on the producer's stream, ending at ring offset `off_end`: an event behind them (lrh_ctx::in_marks; the caller holds mtx_in)
static int input_mark(lrh_ctx *c, long long bytes, int off_end)
{
  lrh_ctx::InMark &m = c->in_marks[c->in_mark_n % 32];
  if (!m.ev) HIPCHK(c, hipEventCreateWithFlags(&m.ev, hipEventDisableTiming));
  c->in_total += bytes; c->in_off_end = off_end;
  HIPCHK(c, hipEventRecord(m.ev, c->stream_in));
  m.total = c->in_total; c->in_mark_n++;
  return LRH_OK;
}
// the noted producer copies (lrh_ctx::in_spans) onto the producer's stream; the caller holds mtx_in
static int flush_input_locked(lrh_ctx *c)
{
  for (const auto &sp : c->in_spans) {
    const int first = sp.nbytes < c->cfg.timf1_bytes - sp.off ? sp.nbytes : c->cfg.timf1_bytes - sp.off;
    HIPCHK(c, hipMemcpyAsync((char *)c->d_timf1 + sp.off, sp.src, first, hipMemcpyHostToDevice, c->stream_in));
    if (sp.nbytes > first) HIPCHK(c, hipMemcpyAsync(c->d_timf1, sp.src + first, sp.nbytes - first, hipMemcpyHostToDevice, c->stream_in));
  }
  if (!c->in_spans.empty()) { const int rc_ = input_mark(c, (long long)c->in_span_bytes, (c->in_spans.back().off + c->in_spans.back().nbytes) & c->timf1_bytemask); if (rc_) return rc_; }
  c->in_spans.clear(); c->in_span_bytes = 0;
  return LRH_OK;
}
int lrh_timf1_write_async(lrh_ctx *c, const void *src, int off, int nbytes)
try {
  if (!c || !src || nbytes < 0 || nbytes > c->cfg.timf1_bytes) return LRH_EINVAL;
  // The workers' noted transforms (lrh_ctx::wparked) and the ones already issued read the ring: a copy that reaches into what they have read
  // since the last such wait -- once per lap of the ring, not once per block -- first has the noted ones issued and goes behind the main stream
  bool guard = false;
  { bool conflict = false;
    { std::lock_guard<std::mutex> lkw(c->mtx_w);
      if (c->rd_any) {
        const long long ring = c->cfg.timf1_bytes, o = off & c->timf1_bytemask, d = ((o - c->rd_lo) % ring + ring) % ring;
        conflict = c->rd_span >= ring || d < c->rd_span || d + nbytes > ring;
        if (conflict) c->rd_any = false;
      } }
    if (conflict) {
      LRH_ENTER(c);
      { const int rc_ = join_handles(c); if (rc_) return rc_; }
      if (!c->ev_in_guard) HIPCHK(c, hipEventCreateWithFlags(&c->ev_in_guard, hipEventDisableTiming));
      HIPCHK(c, hipEventRecord(c->ev_in_guard, c->stream));
      guard = true;
    } }
  // not the context's lock: what this touches besides its own stream are flags the stage calls set or take atomically and events that exist
  // for the life of the context
  std::lock_guard<std::mutex> lk_in(c->mtx_in);
  hipSetDevice(c->cfg.device);
  c->producer_seen = true;
  off &= c->timf1_bytemask;
  const char *s = (const char *)src; char *d = (char *)c->d_timf1;
  const int first = nbytes < c->cfg.timf1_bytes - off ? nbytes : c->cfg.timf1_bytes - off;
  // the fft1 launches already enqueued may still read the ring span being overwritten: the copy goes behind the last of
  // them (the event lrh_fft1_b records), not behind the rest of the chain
  if (c->read_alias_wanted) {
    // another thread is inside lrh_wideband_dsp between its transform and the event that will stand for "timf1 has been read": order the
    // copy behind whatever is on the main stream right now (an event of this path's own, recorded from here)
    if (!c->ev_in_guard) HIPCHK(c, hipEventCreateWithFlags(&c->ev_in_guard, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_in_guard, c->stream)); HIPCHK(c, hipStreamWaitEvent(c->stream_in, c->ev_in_guard, 0));
  } else if (c->fft1_read_valid) HIPCHK(c, hipStreamWaitEvent(c->stream_in, c->ev_fft1_read_cur.load(), 0));
  for (int h = 1; h < LRH_MAX_HANDLES; h++) if (c->hread[h].exchange(false)) HIPCHK(c, hipStreamWaitEvent(c->stream_in, c->hev[h], 0));   // workers that launch themselves (LRH_WORKER_FAST=0, the four-step sizes)
  if (guard) HIPCHK(c, hipStreamWaitEvent(c->stream_in, c->ev_in_guard, 0));
  if (host_span_registered(c, s, (size_t)nbytes)) {         // the caller's page-locked arena (lrh_host_register): the copy engine reads it while the caller goes on
    if (c->in_merge_bytes) {                                 // noted; issued with its neighbours (lrh_ctx::in_spans)
      if (!c->in_spans.empty() && c->in_spans.back().src + c->in_spans.back().nbytes == s && c->in_spans.back().off + c->in_spans.back().nbytes == off &&
          off + nbytes <= c->cfg.timf1_bytes) c->in_spans.back().nbytes += nbytes;
      else c->in_spans.push_back({s, off, nbytes});
      c->in_span_bytes += (size_t)nbytes;
      if (c->in_span_bytes >= c->in_merge_bytes) { const int rc_ = flush_input_locked(c); if (rc_) return rc_; }
    } else {
      HIPCHK(c, hipMemcpyAsync(d + off, s, first, hipMemcpyHostToDevice, c->stream_in));
      if (nbytes > first) HIPCHK(c, hipMemcpyAsync(d, s + first, nbytes - first, hipMemcpyHostToDevice, c->stream_in));
      { const int rc_ = input_mark(c, nbytes, (off + nbytes) & c->timf1_bytemask); if (rc_) return rc_; }
    }
  } else {                                                   // pageable memory: through the producer's own staging buffer, done when this returns (the order on stream_in is the same)
    { const int rc_ = flush_input_locked(c); if (rc_) return rc_; }
    if (!c->h_stage_in && lrh_host_malloc(&c->h_stage_in, LRH_STAGE_BYTES) != hipSuccess) { c->h_stage_in = nullptr; return fail(c, LRH_ENOMEM, "hipHostMalloc(producer staging)"); }
    HIPCHK(c, stage_h2d(c, d + off, s, (size_t)first, c->stream_in, c->h_stage_in));
    if (nbytes > first) HIPCHK(c, stage_h2d(c, d, s + first, (size_t)(nbytes - first), c->stream_in, c->h_stage_in));
    { const int rc_ = input_mark(c, nbytes, (off + nbytes) & c->timf1_bytemask); if (rc_) return rc_; }
  }
  c->in_pending = true;                                      // (the reader records the event behind the copies it needs: wait_for_input)
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_timf1_write_wait(lrh_ctx *c)
try {
  if (!c) return LRH_EINVAL;
  std::lock_guard<std::mutex> lk_in(c->mtx_in);
  hipSetDevice(c->cfg.device);
  { const int rc_ = flush_input_locked(c); if (rc_) return rc_; }
  if (c->stream_in) HIPCHK(c, hipStreamSynchronize(c->stream_in));
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_host_register(lrh_ctx *c, void *ptr, size_t bytes)
try {
  if (!c || !ptr || !bytes) return LRH_EINVAL;
  LRH_ENTER(c);
  HIPCHK(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  { std::lock_guard<std::mutex> lk_in(c->mtx_in); c->host_regs.push_back({(char *)ptr, bytes}); }   // (the producer reads the list under its own lock)
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_host_unregister(lrh_ctx *c, void *ptr)
try {
  if (!c || !ptr) return LRH_EINVAL;
  LRH_ENTER(c);
  // read-backs still on their way into the span first (the copy engine writes there)
  for (int i = 0; i < LRH_NOUT; i++) if (c->out_busy[i] && c->ev_out_done[i]) hipEventSynchronize(c->ev_out_done[i]);
  { std::lock_guard<std::mutex> lk_in(c->mtx_in);
    // producer copies out of the span: the noted ones issued, all of them done before the pages are let go
    { const int rc_ = flush_input_locked(c); if (rc_) return rc_; }
    if (c->stream_in) HIPCHK(c, hipStreamSynchronize(c->stream_in));
    for (size_t i = 0; i < c->host_regs.size(); i++) if (c->host_regs[i].first == (char *)ptr) { c->host_regs.erase(c->host_regs.begin() + i); break; } }
  HIPCHK(c, hipHostUnregister(ptr));
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_timf1_write_packed18(lrh_ctx *c, const void *src, int off, int packed_bytes)
try {
  LRH_ENTER(c);
  if (!c || !src || packed_bytes < 0 || packed_bytes % 9 || (off & 15)) return LRH_EINVAL;
  if (!c->cfg.timf1_dword_input) return fail(c, LRH_ESTATE, "timf1_write_packed18 needs timf1_dword_input");
  if ((long long)packed_bytes / 9 * 16 > c->cfg.timf1_bytes) return LRH_EINVAL;
  if (!packed_bytes) return LRH_OK;
  if ((size_t)packed_bytes > c->pack18_cap) {              // staging buffer for the packed bytes, grown on demand
    if (c->d_pack18) lrh_dev_free(c->d_pack18);
    c->d_pack18 = nullptr; c->pack18_cap = 0;
    if (lrh_dev_malloc((void **)&c->d_pack18, packed_bytes) != hipSuccess) return fail(c, LRH_ENOMEM, "lrh_dev_malloc(packed18 staging)");
    c->pack18_cap = packed_bytes;
  }
  HIPCHK(c, stage_h2d(c, c->d_pack18, src, packed_bytes, c->stream));
  HIPCHK(c, launch_expand18(c->d_pack18, packed_bytes / 9, c->d_timf1, (off & c->timf1_bytemask) / 16, c->cfg.timf1_bytes / 16 - 1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));      // the caller may reuse src
  return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- fft1
// Transforms launched by fft1_b workers on their own streams: whoever reads fft1_float next on the main stream waits for them.
static int launch_parked_fft1(lrh_ctx *c);
static int wait_for_input(lrh_ctx *c, hipStream_t S, int need_end = -1);
static int timf1_read_end(const lrh_ctx *c, int timf1p_ref, int batch);
static Fft1Args fft1_args_of(lrh_ctx *c, int timf1p_ref, int fft1_pa, int batch);
// the blocks the fft1_b workers have noted (lrh_ctx::wparked): in ring order, one launch per contiguous run, on the main stream
static int issue_parked_workers(lrh_ctx *c)
{
  std::vector<lrh_ctx::ParkedW> v;
  { std::lock_guard<std::mutex> lkw(c->mtx_w); v.swap(c->wparked); }
  if (v.empty()) return LRH_OK;
  const int mask = c->fft1n_mask, blk = 2 * c->N1;
  if (c->w_next_nb < 0) {                                    // first ever: the oldest is the one no other entry leads up to
    c->w_next_nb = (v[0].fft1_pa / blk) & mask;
    for (bool moved = true; moved;) { moved = false; for (auto &e : v) if ((((e.fft1_pa / blk) + e.batch) & mask) == c->w_next_nb) { c->w_next_nb = (e.fft1_pa / blk) & mask; moved = true; } }
  }
  const int base = c->w_next_nb;
  std::sort(v.begin(), v.end(), [&](const lrh_ctx::ParkedW &x, const lrh_ctx::ParkedW &y) { return (((x.fft1_pa / blk) - base) & mask) < (((y.fft1_pa / blk) - base) & mask); });
  const long long C = c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1, esz = c->cfg.timf1_dword_input ? 8 : 4;
  hipStream_t keep = c->cur; c->cur = c->stream;
  struct CurBack { lrh_ctx *c; hipStream_t s; ~CurBack() { c->cur = s; } } cur_back{c, keep};
  if (c->in_pending) {                                       // (sorted in ring order: the last entry reads furthest)
    const int rc_ = wait_for_input(c, c->cur, timf1_read_end(c, v.back().timf1p_ref, v.back().batch)); if (rc_) return rc_; }
  size_t i = 0;
  while (i < v.size()) {
    int batch = v[i].batch; size_t j = i + 1;
    while (j < v.size() && batch + v[j].batch <= c->cfg.max_fft1n / 2 &&
           ((v[j].fft1_pa / blk) & mask) == (((v[i].fft1_pa / blk) + batch) & mask) &&
           (v[j].timf1p_ref & c->timf1_bytemask) == (int)(((long long)v[i].timf1p_ref + (long long)batch * c->M1 * esz * C) & c->timf1_bytemask)) { batch += v[j].batch; j++; }
    const Fft1Args a = fft1_args_of(c, v[i].timf1p_ref, v[i].fft1_pa, batch);
    HIPCHK(c, launch_fft1(c->cfg.fft1_n, a, batch, c->cur));
    if (a.real) {                                             // fft1_reherm_dit_one, second half (fft1_re.c:96-131)
      RealSplitArgs r;
      r.spec = c->d_fft1; r.first_nb = a.first_nb; r.nb_mask = a.nb_mask; r.n = c->N1; r.filtercorr = c->d_filtercorr; r.direction = c->cfg.fft1_direction;
      HIPCHK(c, launch_realsplit(r, batch, c->cur));
    } else if (c->d_foldcorr) {
      FoldcorrArgs f;
      f.spec = c->d_fft1; f.first_nb = a.first_nb; f.nb_mask = a.nb_mask; f.n = c->N1;
      f.foldcorr = c->d_foldcorr; f.filtercorr = c->d_filtercorr; f.direction = c->cfg.fft1_direction;
      HIPCHK(c, launch_foldcorr(f, batch, c->cur));
    }
    c->w_next_nb = (a.first_nb + batch) & mask;
    i = j;
  }
  return LRH_OK;
}
static int join_handles(lrh_ctx *c)
{
  if (c->f1_have) { const int rc_ = launch_parked_fft1(c); if (rc_) return rc_; }   // a reader other than the fused make_timf2: the transform after all
  { const int rc_ = issue_parked_workers(c); if (rc_) return rc_; }
  for (int h = 1; h < LRH_MAX_HANDLES; h++)
    if (c->hpending[h]) { HIPCHK(c, hipStreamWaitEvent(c->cur, c->hev[h], 0)); c->hpending[h] = false; }
  return LRH_OK;
}

static Fft1Args fft1_args_of(lrh_ctx *c, int timf1p_ref, int fft1_pa, int batch)
{
  Fft1Args a;
  const int C = c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1;
  const int esz = c->cfg.timf1_dword_input ? 8 : 4;       // bytes per complex sample (fft1.c:420 / :526)
  a.timf1 = c->d_timf1; a.ring_mask = c->cfg.timf1_bytes / esz - 1; a.dword = c->cfg.timf1_dword_input != 0;
  a.shift_i = c->cfg.sample_shift > 0 ? -c->cfg.sample_shift : 0;   // fft1.c:478-482
  a.shift_q = c->cfg.sample_shift < 0 ? c->cfg.sample_shift : 0;    // fft1.c:472-476
  a.chan_count = C; a.chan_index = C > 1 ? c->cfg.timf1_channel_index : 0;
  // first sample of the transform, per channel: ref/2 - 2*C*I1 shorts (fft1.c:421-426; 2-ch: fft1.c:2052-2055)
  a.p0_first = ((timf1p_ref & c->timf1_bytemask) / (esz * C) - c->I1) & (a.ring_mask / C);
  a.step = c->M1; a.window = c->d_window1; a.filtercorr = c->d_filtercorr; a.tw = c->d_tw1; a.out = c->d_fft1;
  a.first_nb = (fft1_pa / (2 * c->N1)) & c->fft1n_mask; a.nb_mask = c->fft1n_mask; a.direction = c->cfg.fft1_direction;
  a.xcd = c->xcd_mask & 1; a.batch = batch;
  a.spare_cus = c->wl_on ? c->spare_cus : 0;                // the limiter inside lrh_wideband_dsp: leave it a compute unit per XCD to start on
  a.real = c->cfg.timf1_real_input != 0;
  if (c->d_foldcorr || a.real) { a.filtercorr = c->d_unitcorr; a.direction = 1; }   // bare transform: k_foldcorr / k_realsplit does the rest
  a.stamps = nullptr;
  return a;
}

// a worker's transforms read [timf1p_ref - I1, + N1 + (batch-1) M1 samples) of the ring: note it for the producer (caller holds mtx_w)
static void worker_read_note(lrh_ctx *c, int timf1p_ref, int batch)
{
  const long long C = c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1, esz = c->cfg.timf1_dword_input ? 8 : 4, ring = c->cfg.timf1_bytes;
  const long long start = (((long long)(timf1p_ref & c->timf1_bytemask) - c->I1 * esz * C) % ring + ring) % ring, len = ((long long)c->N1 + (long long)(batch - 1) * c->M1) * esz * C;
  if (!c->rd_any) { c->rd_any = true; c->rd_lo = start; c->rd_span = len; return; }
  const long long d = ((start - c->rd_lo) % ring + ring) % ring;
  // (a call that starts before the newest end is an earlier block dispatched to a slower worker: inside the span already)
  if (d + len > c->rd_span) c->rd_span = d + len;
}

// ring offset (bytes) one past the last sample the transforms of [timf1p_ref, batch) read (fft1.c:421-426)
static int timf1_read_end(const lrh_ctx *c, int timf1p_ref, int batch)
{
  const long long C = c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1, esz = c->cfg.timf1_dword_input ? 8 : 4, ring = c->cfg.timf1_bytes;
  const long long start = (((long long)(timf1p_ref & c->timf1_bytemask) - c->I1 * esz * C) % ring + ring) % ring;
  return (int)((start + ((long long)c->N1 + (long long)(batch - 1) * c->M1) * esz * C) % ring);
}
// the producer's copies (lrh_timf1_write_async, its own stream) in front of a reader on stream S.  need_end >= 0: the ring offset the reader's samples
// end at -- it waits for the first issue that reaches it (lrh_ctx::in_marks); -1, or no mark to tell: for everything issued so far (an event recorded
// here, by the reader).  The flag that sends readers here goes down only when this reader has waited for the newest issue.
static int wait_for_input(lrh_ctx *c, hipStream_t S, int need_end)
{
  std::lock_guard<std::mutex> lk_in(c->mtx_in);
  if (!c->in_spans.empty()) { const int rc_ = flush_input_locked(c); if (rc_) return rc_; }   // (what the producer has noted but not issued yet)
  if (need_end >= 0 && c->in_mark_n > 0) {
    const long long ring = c->cfg.timf1_bytes, ahead = ((c->in_off_end - need_end) % ring + ring) % ring;   // bytes issued beyond the reader's last
    if (ahead <= ring / 2) {
      const long long need_total = c->in_total - ahead;
      const unsigned kept = c->in_mark_n < 32 ? c->in_mark_n : 32;
      for (unsigned k = c->in_mark_n - kept; k != c->in_mark_n; k++) {
        const lrh_ctx::InMark &m = c->in_marks[k % 32];
        if (m.total < need_total) continue;
        HIPCHK(c, hipStreamWaitEvent(S, m.ev, 0));
        if (k + 1 == c->in_mark_n) c->in_pending = false;
        return LRH_OK;
      }
    }
  }
  std::lock_guard<std::mutex> lk(c->mtx_evin);
  HIPCHK(c, hipEventRecord(c->ev_in, c->stream_in));
  HIPCHK(c, hipStreamWaitEvent(S, c->ev_in, 0));
  c->in_pending = false;
  return LRH_OK;
}

int lrh_fft1_b(lrh_ctx *c, int handle, int timf1p_ref, int fft1_pa, int batch)
try {
  if (c && handle > 0 && handle < LRH_MAX_HANDLES && batch >= 1 && batch <= c->cfg.max_batch && c->worker_fast &&
      !c->corr_on && !c->fft1_big && !c->prof && !c->dbg_stamp && !c->in_dsp && !c->rec) {
    // a worker's call: noted, issued by the next reader of fft1_float (lrh_ctx::wparked)
    std::lock_guard<std::mutex> lkw(c->mtx_w);
    c->wparked.push_back({timf1p_ref, fft1_pa, batch});
    worker_read_note(c, timf1p_ref, batch);
    c->f1_end_valid = false;                                 // worker handles finish in any order
    return LRH_OK;
  }
  LRH_ENTER(c);
  if (!c || batch < 1 || batch > c->cfg.max_batch || handle < 0 || handle >= LRH_MAX_HANDLES) return LRH_EINVAL;
  LRH_WRITES(c, RB(LRH_RING_FFT1_FLOAT));
  // handle 0: the caller's own thread (no_of_fft1b == 0, wcw.c:1036), on the main stream.  handle h >= 1: worker THREAD_FFT1Bh
  // (wcw.c:476-500), on its own stream -- behind everything already enqueued on the main stream (the earlier readers of the
  // ring slots it overwrites, the previous lap) and ahead of the next reader of fft1_float (join_handles).
  hipStream_t const keep_cur = c->cur;
  struct CurBack { lrh_ctx *c; hipStream_t s; ~CurBack() { c->cur = s; } } cur_back{c, keep_cur};
  if (handle > 0) {
    if (c->rec) return fail(c, LRH_ESTATE, "fft1_b workers cannot run inside lrh_wideband_dsp");
    if (!c->hstream[handle]) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->hstream[handle], hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&c->hev[handle], hipEventDisableTiming));
      if (!c->hev_start) HIPCHK(c, hipEventCreateWithFlags(&c->hev_start, hipEventDisableTiming));
    }
    HIPCHK(c, hipEventRecord(c->hev_start, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->hstream[handle], c->hev_start, 0));
    if (c->in_pending) { const int rc_ = wait_for_input(c, c->hstream[handle], timf1_read_end(c, timf1p_ref, batch)); if (rc_) return rc_; }      // every worker waits for the producer's copy
    { std::lock_guard<std::mutex> lkw(c->mtx_w); worker_read_note(c, timf1p_ref, batch); }
    c->cur = c->hstream[handle];
  } else if (c->in_pending) { const int rc_ = wait_for_input(c, c->cur, timf1_read_end(c, timf1p_ref, batch)); if (rc_) return rc_; }   // samples of lrh_timf1_write_async
  Fft1Args a = fft1_args_of(c, timf1p_ref, fft1_pa, batch);
  const int C = a.chan_count;
  if (c->dbg_stamp) {                                     // diagnostics (LRH_STAMP=1, -DLRH_STAMP_BUILD): phase stamps of this launch to stderr
    if (!c->d_stamps) HIPCHK(c, lrh_dev_malloc((void **)&c->d_stamps, 2 * LRH_STAMPS_PER_WG * sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(c->d_stamps, 0, 2 * LRH_STAMPS_PER_WG * sizeof(unsigned long long), c->cur));
    a.stamps = c->d_stamps;
  }
  // k_fft1w: fft1_size 16384, int16; k_fft1v: 4096 / 8192 / 16384, int16 or int32
  if (c->f1_defer && handle == 0 && (c->fuse_v ? (c->cfg.fft1_n >= 12 && c->cfg.fft1_n <= 14) : (c->cfg.fft1_n == 14 && !a.dword)) && (!a.real || (c->fuse_v && c->fuse_real && c->cfg.fft1_direction > 0 && a.chan_count == 1)) && !a.shift_i && !a.shift_q &&
      !c->d_foldcorr && a.direction > 0 && !c->dbg_stamp) {
    if (c->f1_have) { const int rc_ = launch_parked_fft1(c); if (rc_) return rc_; }
    c->f1_cont = c->f1_end_valid && a.p0_first == c->f1_end;
    c->f1_end = (a.p0_first + batch * a.step) & (a.ring_mask / C); c->f1_end_valid = true;
    c->f1_args = a; c->f1_batch = batch; c->f1_have = true;    // lrh_make_timf2 takes it from here (k_fft1w)
    return LRH_OK;
  }
  if (handle == 0) { c->f1_end = (a.p0_first + batch * a.step) & (a.ring_mask / C); c->f1_end_valid = true; }
  else c->f1_end_valid = false;                                // worker handles finish in any order
  const bool defer_big = c->f1_defer && c->fft1_big && handle == 0 && c->cfg.fft1_n == 15 && !a.real && !a.shift_i && !a.shift_q && !c->d_foldcorr && a.direction > 0 && !c->dbg_stamp;
  if (defer_big && c->f1_have) { const int rc_ = launch_parked_fft1(c); if (rc_) return rc_; }
  ProfScope ps(c, "fft1");
  if (c->fft1_big) {
    const size_t need = (size_t)batch * c->N1;
    if (c->fft1_scratch_cap[handle] < need) {
      HIPCHK(c, hipStreamSynchronize(c->cur));
      if (c->d_fft1_scratch[handle]) lrh_dev_free(c->d_fft1_scratch[handle]);
      c->d_fft1_scratch[handle] = nullptr; c->fft1_scratch_cap[handle] = 0;
      const int rc_ = dev_alloc(c, &c->d_fft1_scratch[handle], need, false); if (rc_) return rc_;
      c->fft1_scratch_cap[handle] = need;
    }
    Fft1BigArgs g; g.f = a; g.tw_a = c->d_tw1a; g.tw_b = c->d_tw1b; g.tw_big = c->d_tw1; g.scratch = c->d_fft1_scratch[handle];
    if (defer_big) {                                       // column step now, the row step rides in lrh_make_timf2's kernel
      HIPCHK(c, launch_fft1_big(c->cfg.fft1_n, g, batch, c->cur, 1));
      c->f1_big = g; c->f1_args = a; c->f1_batch = batch; c->f1_have = true; c->f1_is_big = true;
      if (c->ev_fft1_read) { HIPCHK(c, hipEventRecord(c->ev_fft1_read, c->cur)); c->ev_fft1_read_cur = c->ev_fft1_read; c->fft1_read_valid = true; }   // timf1 has been read
      return LRH_OK;
    }
    HIPCHK(c, launch_fft1_big(c->cfg.fft1_n, g, batch, c->cur));
  } else
  HIPCHK(c, launch_fft1(c->cfg.fft1_n, a, batch, c->cur));
  if (c->dbg_stamp) {
    unsigned long long h[2 * LRH_STAMPS_PER_WG];
    HIPCHK(c, stage_d2h(c, h, c->d_stamps, sizeof h, c->cur)); HIPCHK(c, hipStreamSynchronize(c->cur));
    for (int w = 0; w < 2; w++) {
      fprintf(stderr, "fft1 stamps wg%d:", w ? 128 : 0);
      for (int i = 1; i < LRH_STAMPS_PER_WG && h[w * LRH_STAMPS_PER_WG + i]; i++) fprintf(stderr, " %llu", h[w * LRH_STAMPS_PER_WG + i] - h[w * LRH_STAMPS_PER_WG]);
      fprintf(stderr, "\n");
    }
  }
  // timf1 has been read: producer copies may follow.  Inside lrh_wideband_dsp's serial order lrh_make_timf2 comes next on this stream and
  // records ev_timf2_done anyway: that event tells the producer, and the transform is not followed by a record of its own (a packet the
  // queue works off before the next kernel: 6 us of a 40 us call of one block)
  if (handle == 0 && c->ev_fft1_read) {
    if (c->in_dsp && c->cfg.second_fft_enable && c->cur == c->stream && !c->rec && !c->producer_seen) c->read_alias_wanted = true;
    else { HIPCHK(c, hipEventRecord(c->ev_fft1_read, c->cur)); c->ev_fft1_read_cur = c->ev_fft1_read; c->fft1_read_valid = true; }
  }
  if (a.real) {                                             // fft1_reherm_dit_one, second half (fft1_re.c:96-131)
    RealSplitArgs r;
    r.spec = c->d_fft1; r.first_nb = a.first_nb; r.nb_mask = a.nb_mask; r.n = c->N1; r.filtercorr = c->d_filtercorr; r.direction = c->cfg.fft1_direction;
    HIPCHK(c, launch_realsplit(r, batch, c->cur));
  } else if (c->d_foldcorr) {
    FoldcorrArgs f;
    f.spec = c->d_fft1; f.first_nb = a.first_nb; f.nb_mask = a.nb_mask; f.n = c->N1;
    f.foldcorr = c->d_foldcorr; f.filtercorr = c->d_filtercorr; f.direction = c->cfg.fft1_direction;
    HIPCHK(c, launch_foldcorr(f, batch, c->cur));
  }
  if (handle > 0) {                                         // the worker's transforms: awaited by the next reader of fft1_float and by the producer
    HIPCHK(c, hipEventRecord(c->hev[handle], c->cur));
    c->hpending[handle] = true; c->hread[handle] = true;
  }
  return LRH_OK;
}
LRH_CATCH(c)

static int launch_parked_fft1(lrh_ctx *c)
{
  c->f1_have = false;
  hipStream_t keep = c->cur; c->cur = c->stream;          // parked on the main stream, issued there
  struct CurBack { lrh_ctx *c; hipStream_t s; ~CurBack() { c->cur = s; } } cur_back{c, keep};
  ProfScope ps(c, "fft1");
  if (c->f1_is_big) { c->f1_is_big = false; HIPCHK(c, launch_fft1_big(c->cfg.fft1_n, c->f1_big, c->f1_batch, c->cur, 2)); return LRH_OK; }   // the row step
  HIPCHK(c, launch_fft1(c->cfg.fft1_n, c->f1_args, c->f1_batch, c->cur));
  if (c->ev_fft1_read) { HIPCHK(c, hipEventRecord(c->ev_fft1_read, c->cur)); c->ev_fft1_read_cur = c->ev_fft1_read; c->fft1_read_valid = true; }
  return LRH_OK;
}

int lrh_set_foldcorr(lrh_ctx *c, const float *fc)
try {
  if (!c) return LRH_EINVAL;
  LRH_ENTER(c);
  { const int rcj_ = join_handles(c); if (rcj_) return rcj_; }
  HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2)); if (c->stream_sel) HIPCHK(c, hipStreamSynchronize(c->stream_sel));
  if (!fc) { if (c->d_foldcorr) lrh_dev_free(c->d_foldcorr); c->d_foldcorr = nullptr; return LRH_OK; }
  if (c->cfg.timf1_real_input) return fail(c, LRH_ESTATE, "no I/Q mirror image with real samples (init_foldcorr is I/Q only, buf.c:1461)");
  const size_t bytes = sizeof(float2) * c->N1;
  if (!c->d_foldcorr && lrh_dev_malloc((void **)&c->d_foldcorr, bytes) != hipSuccess) return fail(c, LRH_ENOMEM, "lrh_dev_malloc(foldcorr)");
  if (!c->d_unitcorr) {
    if (lrh_dev_malloc((void **)&c->d_unitcorr, bytes) != hipSuccess) return fail(c, LRH_ENOMEM, "lrh_dev_malloc(unit filter table)");
    std::vector<float2> one(c->N1, make_float2(1.f, 0.f));
    HIPCHK(c, stage_h2d(c, c->d_unitcorr, one.data(), bytes, c->stream));
  }
  HIPCHK(c, stage_h2d(c, c->d_foldcorr, fc, bytes, c->stream));
  return LRH_OK;
}
LRH_CATCH(c)

// fft1_c: power sums (fft1.c:4115-4171), counters (fft1.c:4507-4523), slow average (fft1.c:4526-4605)
int lrh_fft1_c(lrh_ctx *c, lrh_ptrs *p, int batch)
try {
  LRH_ENTER(c);
  if (!c || !p || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  LRH_WRITES(c, RB(LRH_RING_FFT1_FLOAT) | RB(LRH_RING_FFT1_SUMSQ) | RB(LRH_RING_FFT1_SLOWSUM) | RB(LRH_RING_FFT1_CORRSUM) | RB(LRH_RING_FFT1_SLOWCORR) | RB(LRH_RING_FFT1_SLOWCORR_TOT));
  if (!(c->ss_defer && c->f1_have)) { const int rc_ = join_handles(c); if (rc_) return rc_; }   // parked sums behind a parked transform: make_timf2 decides
  const int N = c->N1, avg1 = c->cfg.fft_avg1num, last = N - 1;
  if ((p->fft1_sumsq_counter + batch + avg1 - 1) / avg1 + c->cfg.fft_avg2num + 1 > c->cfg.fft1_sumsq_bufsize / N)
    return fail(c, LRH_EINVAL, "fft1_sumsq ring too short for this batch");
  if (c->spur_max && !c->cfg.second_fft_enable) {
    // the last step of fft1 when the AFC runs from fft1 (fft1afc_flag > 0, fft1.c:4196-4244, 4428-4460): eliminate_spurs on the new transforms
    // in order, then the search spectrum's rows from their powers -- in front of the sums, which are formed from the cleaned spectra
    if (c->spur_n > 0) { SpurArgs sp; spur_args(c, &sp, p->fft1_nb, batch); ProfScope ps(c, "spur"); HIPCHK(c, launch_spur(sp, c->cur)); }
    for (int b = 0; b < batch; b++) { const int rcs_ = spur_search_row(c, (p->fft1_nb + b) & c->fft1n_mask, c->cur); if (rcs_) return rcs_; }
  }
  SumsqArgs sa;
  sa.spec = c->d_fft1; sa.nb_mask = c->fft1n_mask; sa.n = N; sa.sumsq = c->d_sumsq; sa.sumsq_mask = c->sumsq_mask;
  sa.first_nb = p->fft1_nb; sa.batch = batch; sa.avg = avg1; sa.c0 = p->fft1_sumsq_counter; sa.pa0 = p->fft1_sumsq_pa;
  if (c->ss_defer) {
    if (c->ss_have) {                                      // nobody took the last ones
      if (c->f1_have && c->f1_args.first_nb == c->ss_args.first_nb) { const int rc_ = launch_parked_fft1(c); if (rc_) return rc_; }
      ProfScope ps(c, "sumsq"); HIPCHK(c, launch_sumsq(c->ss_args, c->cur));
    }
    c->ss_args = sa; c->ss_have = true;
  } else { ProfScope ps(c, "sumsq"); HIPCHK(c, launch_sumsq(sa, c->cur)); }
  std::vector<std::function<int(lrh_ctx *)>> *const rec0 = c->rec;
  if (c->ss_defer) c->rec = &c->ss_queue;                // the slow average waits for the sums
  struct RecBack { lrh_ctx *c; std::vector<std::function<int(lrh_ctx *)>> *r; ~RecBack() { c->rec = r; } } rec_back{c, rec0};
  const int nupd = (p->fft1_sumsq_counter + batch) / avg1;              // groups completed by this batch
  if (nupd > 0) {
    SlowsumArgs ua;
    ua.sumsq = c->d_sumsq; ua.slowsum = c->d_slowsum; ua.n = N; ua.bufsize = c->cfg.fft1_sumsq_bufsize; ua.avg2 = c->cfg.fft_avg2num;
    ua.nupd = nupd; ua.pa0 = p->fft1_sumsq_pa; ua.recalc0 = p->fft1_sumsq_recalc; ua.step = c->cfg.wg_xpoints / c->cfg.slowsum_fresh_recalc;
    // one full cycle of the rolling refresh takes last/step (+ wrap) updates: start the kernel's search that far back
    const int cycle = last / (ua.step > 0 ? ua.step : 1) + 3;
    ua.e0 = nupd > cycle ? nupd - cycle : 0; ua.recalc_e0 = ua.recalc0;
    for (int e = 0; e < nupd; e++) {                                     // same recursion as the kernel (fft1.c:4568-4573)
      if (e == ua.e0) ua.recalc_e0 = p->fft1_sumsq_recalc;
      if (p->fft1_sumsq_recalc == last) p->fft1_sumsq_recalc = 0;
      p->fft1_sumsq_recalc += ua.step; if (p->fft1_sumsq_recalc > last) p->fft1_sumsq_recalc = last;
    }
    LRH_DEVICE_WORK(c, { ProfScope ps(c, "slowsum"); HIPCHK(c, launch_slowsum(ua, c->cur)); c->sums_stream = c->cur; });
    if (c->cfg.second_fft_enable) p->fft1_liminfo_cnt += nupd;          // fft1.c:4515-4518
    p->fft1_sumsq_pa = (p->fft1_sumsq_pa + nupd * N) & c->sumsq_mask;
  }
  p->fft1_sumsq_counter = (p->fft1_sumsq_counter + batch) % avg1;
  p->fft1_nb = (p->fft1_nb + batch) & c->fft1n_mask; p->fft1_pb = p->fft1_nb * 2 * N;
  return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- timf2
int lrh_make_timf2(lrh_ctx *c, lrh_ptrs *p, int batch)
try {
  LRH_ENTER(c);
  if (!c || !p || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  // (the sums only when they are parked for this call's kernel: a caller that has just begun to read this period's spectra back --
  // the glue between lrh_fft1_c and this call -- must not have its weak stream queued behind those copies, 110 us per call at fft1_size 16384)
  LRH_WRITES(c, RB(LRH_RING_FFT1_FLOAT) | (c->ss_have ? RB(LRH_RING_FFT1_SUMSQ) | RB(LRH_RING_FFT1_SLOWSUM) : 0u) | RB(LRH_RING_TIMF2_FLOAT) | RB(LRH_RING_TIMF2_PWR));
  // k_fft1w: the parked forward transform, the parked sums and this call's weak stream address the same transforms
  const int nb_here = (p->fft1_px / (2 * c->N1)) & c->fft1n_mask;
  const bool fused_any = c->f1_have && c->ss_have && c->timf2_mode == 1 && c->f1_batch == batch && c->f1_args.first_nb == nb_here &&
                         c->ss_args.batch == batch && c->ss_args.first_nb == nb_here && c->cur == c->stream;
  // a gap in the walk over timf1 (or new tables) since the previous call: the ring no longer holds the partner's input.  With the whole
  // spectrum kept the two-kernel path takes this call (its partner is the previous spectrum in the fft1 ring, the reference's own
  // carry); with the sparse ring the call starts over like the first one of a stream (lrh_fft1_b in include/linrad_hip.h)
  const bool partner_lost = c->timf2_primed && !c->f1_cont;
  const bool keeps_spec = !(c->cfg.fft1_float_sparse && !c->corr_on);
  const bool fused1 = fused_any && !c->f1_is_big && c->d_ss_part && !(partner_lost && keeps_spec);
  bool read_alias = false;
  const bool fused15 = fused_any && c->f1_is_big;          // fft1_size 32768: row step + sums + column step of both streams (k_fft1r_t2c)
  if (!fused1 && !fused15) { const int rc_ = join_handles(c); if (rc_) return rc_; }
  if (c->sel_table_pending) { HIPCHK(c, hipStreamWaitEvent(c->cur, c->ev_sel, 0)); c->sel_table_pending = false; }   // routing words from the side stream
  if (c->h_sel_low) sellim_poll(c);                        // weak-bin count of the newest finished limiter update, if one has arrived
  Timf2Args a;
  a.spec = c->d_fft1; a.first_nb = (p->fft1_px / (2 * c->N1)) & c->fft1n_mask; a.nb_mask = c->fft1n_mask;
  a.pack_cur = c->d_pack_cur; a.pack_prev = c->pack_prev_stale ? c->d_pack_prev : c->d_pack_cur; a.tw = c->d_tw1;
  a.timf2w = c->d_timf2w; a.timf2s = c->d_timf2s; a.pwr = c->d_pwr; a.pa_first = p->timf2_pa / 4; a.mask = c->timf2pow_mask; a.step = c->M1;
  a.mode = c->timf2_mode; a.ia = c->I1 / 2; a.invwin = c->d_invwin1;
  a.ampfac = (float)(1.0 / (1 << c->cfg.bckfft_att_n));
  a.xcd = (c->xcd_mask >> 1) & 1;
  a.spare_cus = c->wl_on ? c->spare_cus : 0;
  auto plain_timf2 = [&]() -> int {
    ProfScope ps(c, "timf2");
    if (!c->fft1_big) { HIPCHK(c, launch_timf2(c->cfg.fft1_n, a, batch, c->cur)); return LRH_OK; }
    const size_t need = (size_t)batch * 2 * c->N1;
    if (c->timf2_scratch_cap < need) {
      HIPCHK(c, hipStreamSynchronize(c->cur));
      if (c->d_timf2_scratch) lrh_dev_free(c->d_timf2_scratch);
      c->d_timf2_scratch = nullptr; c->timf2_scratch_cap = 0;
      const int rc_ = dev_alloc(c, &c->d_timf2_scratch, need, false); if (rc_) return rc_;
      c->timf2_scratch_cap = need;
    }
    Timf2BigArgs g; g.t = a; g.tw_a = c->d_tw1a; g.tw_b = c->d_tw1b; g.tw_big = c->d_tw1; g.scratch = c->d_timf2_scratch;
    HIPCHK(c, launch_timf2_big(c->cfg.fft1_n, g, batch, c->cur));
    return LRH_OK;
  };
  if (fused15) {
    const size_t need = (size_t)batch * 2 * c->N1;
    if (c->timf2_scratch_cap < need) {
      HIPCHK(c, hipStreamSynchronize(c->cur));
      if (c->d_timf2_scratch) lrh_dev_free(c->d_timf2_scratch);
      c->d_timf2_scratch = nullptr; c->timf2_scratch_cap = 0;
      const int rc_ = dev_alloc(c, &c->d_timf2_scratch, need, false); if (rc_) return rc_;
      c->timf2_scratch_cap = need;
    }
    Fft1rT2cArgs w; w.f1 = c->f1_big; w.ss = c->ss_args; w.groups_per_run = 0; w.keep_spec = (c->cfg.fft1_float_sparse && !c->corr_on) ? 0 : 1;
    w.t2.t = a; w.t2.tw_a = c->d_tw1a; w.t2.tw_b = c->d_tw1b; w.t2.tw_big = c->d_tw1; w.t2.scratch = c->d_timf2_scratch;
    c->ss_have = false; c->f1_have = false; c->f1_is_big = false;
    { ProfScope ps(c, "fft1w"); HIPCHK(c, launch_fft1r_t2c(w, batch, c->cur)); }
  } else if (fused1) {
    const SumsqArgs sa = c->ss_args;
    const Fft1Args &f = c->f1_args;
    c->ss_have = false; c->f1_have = false;
    float *const part = c->d_ss_part + (size_t)c->ss_flip * c->ss_part_stride; c->ss_flip ^= 1;
    Fft1wArgs w; memset(&w, 0, sizeof w);
    w.timf1 = f.timf1; w.ring_mask = f.ring_mask; w.p0_first = f.p0_first; w.step = f.step; w.chan_count = f.chan_count; w.chan_index = f.chan_index;
    w.window = f.window; w.filtercorr = f.filtercorr; w.tw = f.tw;
    w.spec = c->d_fft1; w.first_nb = a.first_nb; w.nb_mask = a.nb_mask; w.keep_spec = (c->cfg.fft1_float_sparse && !c->corr_on) ? 0 : 1;
    w.pack_cur = a.pack_cur; w.pack_prev = a.pack_prev; w.timf2w = a.timf2w; w.pwr = a.pwr; w.pa_first = a.pa_first; w.mask = a.mask; w.ampfac = a.ampfac;
    w.have_prev = (c->timf2_primed && !partner_lost) ? 1 : 0;
    w.ss_ring = sa.sumsq; w.ss_part = part; w.ss_mask = sa.sumsq_mask; w.ss_avg = sa.avg; w.ss_c0 = sa.c0; w.ss_pa0 = sa.pa0;
    w.batch = batch; w.spare_cus = a.spare_cus;
    w.filtercorr_v = c->d_filtercorr_v; w.max_wg = (int)(c->ss_part_stride / (2 * (size_t)c->N1));
    static const bool v_stamps = getenv("LRH_FFT1V_EXP") && (atoi(getenv("LRH_FFT1V_EXP")) & 2);
    if (v_stamps) {
      if (!c->d_stamps) HIPCHK(c, lrh_dev_malloc((void **)&c->d_stamps, 64 * sizeof(unsigned long long)));
      HIPCHK(c, hipMemsetAsync(c->d_stamps, 0, 64 * sizeof(unsigned long long), c->cur));
      w.stamps = c->d_stamps;
    }
    { ProfScope ps(c, "fft1w");
      w.real_peak = f.real ? c->h_window1_ref[c->N1] : 0.f;
      if (c->fuse_v) HIPCHK(c, launch_fft1v(c->cfg.fft1_n, f.dword != 0, f.real != 0, w, c->cur, &c->ss_run));
      else HIPCHK(c, launch_fft1w(w, c->cur, &c->ss_run)); }
    if (v_stamps) {
      static int printed = 0;
      unsigned long long h[64];
      HIPCHK(c, stage_d2h(c, h, c->d_stamps, sizeof h, c->cur)); HIPCHK(c, hipStreamSynchronize(c->cur));
      if (printed++ < 3) for (int wv = 0; wv < 2; wv++) {
        fprintf(stderr, "fft1v stamps wave%d:", wv ? 4 : 0);
        for (int i = 1; i < 32 && h[wv * 32 + i]; i++) fprintf(stderr, " %llu", h[wv * 32 + i] - h[wv * 32]);
        fprintf(stderr, "\n");
      }
    }
    // timf1 has been read: ev_timf2_done, recorded below behind the strong stream's kernel, tells the producer (one record less between the two kernels)
    // (with a streaming producer -- lrh_timf1_write_async has been called -- the record stays where the ring has been read: a copy-bound
    // caller gets the ring back 140 us earlier per round, 9.2 -> 10.6 Gsamples/s over PCIe, for one more packet on the main stream)
    if (c->ev_fft1_read && c->producer_seen) { HIPCHK(c, hipEventRecord(c->ev_fft1_read, c->cur)); c->ev_fft1_read_cur = c->ev_fft1_read; c->fft1_read_valid = true; }
    else read_alias = c->ev_fft1_read != nullptr;
    { ProfScope ps(c, "timf2s"); HIPCHK(c, launch_timf2_strong(c->cfg.fft1_n, a, batch, c->cur)); }
    const SumsqArgs ja = sa; const int run = c->ss_run;
    c->ss_queue.insert(c->ss_queue.begin(), [ja, run, part](lrh_ctx *c) -> int {
      ProfScope ps(c, "sumsq_join"); HIPCHK(c, launch_sumsq_join(ja, part, run, c->cur)); return LRH_OK; });
  } else if (c->ss_have) {
    const SumsqArgs &sa = c->ss_args;
    c->ss_have = false;
    if (!c->fft1_big && a.mode == 1 && c->d_ss_part && sa.batch == batch && sa.first_nb == a.first_nb) {
      // the scratch of split groups alternates between two halves: in the two-stream schedules the join of this launch runs
      // on the side stream and may still be reading when the next launch starts writing
      float *const part = c->d_ss_part + (size_t)c->ss_flip * c->ss_part_stride; c->ss_flip ^= 1;
      { ProfScope ps(c, "timf2"); HIPCHK(c, launch_timf2(c->cfg.fft1_n, a, batch, c->cur, &sa, part, &c->ss_run)); }
      const SumsqArgs ja = sa; const int run = c->ss_run;
      if (run > 0) c->ss_queue.insert(c->ss_queue.begin(), [ja, run, part](lrh_ctx *c) -> int {      // (0: one workgroup wrote the ring itself)
        ProfScope ps(c, "sumsq_join"); HIPCHK(c, launch_sumsq_join(ja, part, run, c->cur)); return LRH_OK; });
    } else {                                             // pointers out of step: separate pass after all
      { ProfScope ps(c, "sumsq"); HIPCHK(c, launch_sumsq(sa, c->cur)); }
      { const int rc_ = plain_timf2(); if (rc_) return rc_; }
    }
  } else { const int rc_ = plain_timf2(); if (rc_) return rc_; }
  // from now on the previous transform was routed with the current table
  c->pack_prev_stale = false;                            // (the table in d_pack_cur is the previous transform's from here on: no copy)
  HIPCHK(c, hipEventRecord(c->ev_timf2_done, c->cur)); c->timf2_done_valid = true;
  c->stage_valid[LRH_STAGE_TIMF2] = true;                // lrh_stage_wait(LRH_STAGE_TIMF2) waits on ev_timf2_done itself: no second record
  if (!c->in_dsp && !c->rec) { const int rcr_ = stage_ring_mark(c, LRH_STAGE_TIMF2); if (rcr_) return rcr_; }
  if (read_alias || c->read_alias_wanted) { c->ev_fft1_read_cur = c->ev_timf2_done; c->fft1_read_valid = true; c->read_alias_wanted = false; }
  c->timf2_primed = true;
  const int low = c->lowlevel_points;
  for (int b = 0; b < batch; b++) {                                    // timf2.c:127-128, 205-207
    p->fft1_px = (p->fft1_px + 2 * c->N1) & c->fft1_mask;
    p->fft1_nx = (p->fft1_nx + 1) & c->fft1n_mask;
    p->fft1_lowlevel_points = low;
    p->fft1_lowlevel_fraction = 0.02 * (49 * p->fft1_lowlevel_fraction + low / ((float)(c->N1 - 1)));
    p->timf2_pa = (p->timf2_pa + 4 * c->M1) & c->timf2_mask;
  }
  return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- blanker
// second half of a blanker call: what follows the pulse search -- bookkeeping that needs the search's resume point, statistics, dumb blanker
static int blanker_tail(lrh_ctx *c, lrh_ptrs *p, BlankArgs a, int pbeg, const int *out, float lowlevel_fraction, bool coupled)
{
  const int mask = c->timf2pow_mask;
  if (out) {
    p->timf2p_fit = (out[0] - 16 + mask) & (mask & ~3);                  // blank1.c:1458-1461
    a.post_stats = 1; a.fitted = out[1]; a.rejected = out[2]; a.clever_mode = c->bt.clever_bln_mode; a.clever_factor = c->bt.clever_bln_factor;
  }
  const int m = (p->timf2p_fit - pbeg + 1 + mask) & mask;
  p->timf2_blanker_points += m;
  a.m = m; a.nstat = (out ? ((p->timf2p_fit - pbeg) & mask) : a.total) / 4; a.blanker_points = p->timf2_blanker_points;
  a.npartials = a.nstat < 4096 ? 1 : (a.nstat / 4096 < LRH_BLN_PARTIALS ? a.nstat / 4096 : LRH_BLN_PARTIALS);
  a.interval = c->cfg.blanker_info_update_interval; a.avgnum = c->cfg.timf2_noise_floor_avgnum; a.factor = c->cfg.stupid_bln_factor;
  a.lowlevel_fraction = lowlevel_fraction;
  p->blanker_info_update_counter++;                                      // blank1.c:1550-1601
  a.do_update = 0;
  a.debug = c->dbg_bln;
  if (p->blanker_info_update_counter >= a.interval) {
    if (lowlevel_fraction < 0.1) p->blanker_info_update_counter--;
    else { a.do_update = 1; p->blanker_info_update_counter = 0; p->timf2_blanker_points = 0; }
  }
  const int ring_words = c->cfg.timf2pow_size / 32;
  LRH_DEVICE_WORK(c, { ProfScope ps(c, "blanker"); HIPCHK(c, launch_blanker(a, ring_words, c->cur)); });
  if (coupled) { c->fin_args = a; c->fin_args.phase = 2; c->fin_pending = true; }
  return LRH_OK;
}
// the search of the previous (deferred) call has been issued by now: wait for its resume point and issue the rest of that call on the side stream
static int clever_late_finish(lrh_ctx *c, lrh_ptrs *p)
{
  if (!c->clv_issued) return fail(c, LRH_ESTATE, "linear blanker: the search of the previous round has not been issued");
  HIPCHK(c, hipEventSynchronize(c->ev_clv));
  c->clv_wait = false; c->clv_issued = false;
  if (c->h_clv_out[3]) {                                    // colliding extents: the one-wave replay, now, and its result
    HIPCHK(c, launch_clever(c->clv_late.ca, c->stream2, 4));
    HIPCHK(c, hipMemcpyAsync(c->h_clv_out, (char *)c->d_bst + offsetof(BlankState, clever_out), 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream2));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
  }
  const int out[3] = { c->h_clv_out[0], c->h_clv_out[1], c->h_clv_out[2] };
  std::vector<std::function<int(lrh_ctx *)>> *keep_rec = c->rec; hipStream_t keep_cur = c->cur;
  c->rec = nullptr; c->cur = c->stream2;
  int rc = blanker_tail(c, p, c->clv_late.a, c->clv_late.pbeg, out, c->clv_late.lowlevel, false);
  if (!rc && hipEventRecord(c->ev_blank, c->stream2) != hipSuccess) rc = fail(c, LRH_EDEVICE, "hipEventRecord");    // fft2 of that round waits for this
  c->rec = keep_rec; c->cur = keep_cur;
  return rc;
}
int lrh_first_noise_blanker(lrh_ctx *c, lrh_ptrs *p)
try {
  LRH_ENTER(c);
  if (!c || !p) return LRH_EINVAL;
  LRH_WRITES(c, RB(LRH_RING_TIMF2_FLOAT) | RB(LRH_RING_TIMF2_PWR));
  if (c->clv_wait) { const int rc_ = clever_late_finish(c, p); if (rc_) return rc_; }    // this call starts where that search stopped
  const int mask = c->timf2pow_mask;
  const int pbeg = p->timf2p_fit;
  int pend = (p->timf2_pa / 4 - c->cfg.blnfit_range + mask) & mask;     // blank1.c:705-710
  pend &= 0xfffffffc;
  const int total = (pend - pbeg + 1 + mask) & mask;
  if (total < c->cfg.blanker_min_points) return LRH_OK;                  // rate limit, blank1.c:712-715
  const bool coupled = c->cfg.blanker_channels == 2;
  BlankArgs a; memset(&a, 0, sizeof a);
  a.pwr = c->d_pwr; a.timf2w = c->d_timf2w; a.mask_bits = c->d_blnbits; a.mask = mask;
  a.pbeg = pbeg; a.total = (pend - pbeg) & mask;
  a.chans = 1;
  if (coupled) {
    if (c->fin_pending) return fail(c, LRH_ESTATE, "lrh_blanker_finish of the previous call is missing");
    if (c->x_pbeg != pbeg || c->x_count < 0 || c->x_span != a.total) return fail(c, LRH_ESTATE, "lrh_blanker_begin was not called for this span");
    { ProfScope ps(c, "xcopy"); HIPCHK(c, launch_span_copy(c->d_xbuf, c->d_pwr_sum, pbeg, c->x_count, mask, 1, c->cur)); }   // the exchanged sums take their ring places
    if (c->clever_on) {                                                   // and so do the partner channel's samples
      if (c->xw_count <= 0) return fail(c, LRH_ESTATE, "lrh_blanker_begin ran before the blanker tables were installed");
      const int nw = c->xw_count / 2;
      HIPCHK(c, launch_span_copy2(c->d_xweak + (size_t)(1 - (c->cfg.timf1_channel_index & 1)) * nw, c->d_tf_partner, pbeg - c->cfg.blnfit_range, nw, mask, 1, c->cur));
    }
    c->x_count = -1;
    a.pwr = c->d_pwr_sum; a.own = c->d_pwr; a.xstat = c->d_xstat; a.own_slot = c->cfg.timf1_channel_index & 1; a.chans = 2; a.phase = 1;
  }
  a.clr1 = (c->cfg.blanker_pulsewidth + 1) >> 1; a.clr2 = c->cfg.blanker_pulsewidth + 1;     // blank1.c:1013-1014
  a.mode = c->cfg.stupid_bln_mode; a.st = c->d_bst; a.partials = c->d_partials; a.tiles = c->d_bln_tiles; a.counts = c->d_bln_counts;
  a.wbusy = c->d_bln_wbusy; a.wstate = c->d_bln_wstate;
  p->timf2p_fit = pend; p->timf2_pn2 = 4 * pend;                         // blank1.c:1464-1466
  if (c->clever_on) {
    // the pulse search runs first (blank1.c:765-1003) and decides where the next call resumes: one int comes back, the call waits for it
    if (c->rec && coupled) return fail(c, LRH_ESTATE, "linear blanker of two coupled channels inside the deferred schedule");
    if (a.total > c->cfg.timf2pow_size - 1024) return fail(c, LRH_EINVAL, "linear blanker: span longer than the timf2 power ring");   // the backup keeps 256 samples either side
    CleverArgs ca; memset(&ca, 0, sizeof ca);
    ca.pwr = coupled ? c->d_pwr_sum : c->d_pwr; ca.timf2w = c->d_timf2w; ca.flag = c->d_bln_flag; ca.cand = c->d_bln_cand; ca.mask = mask;
    if (coupled) { ca.twochan = 1; ca.chan = c->cfg.timf1_channel_index & 1; ca.timf2y = c->d_tf_partner; ca.pwr_own = c->d_pwr; }
    ca.pbeg = pbeg; ca.total = a.total; ca.R = c->cfg.blnfit_range; ca.pwid = c->cfg.blanker_pulsewidth; ca.rs = c->bt.refpul_size;
    ca.largest = c->bt.largest_blnfit; ca.amp_factor = c->bt.liminfo_amplitude_factor;
    ca.refpulse = c->d_bt_refpulse; ca.phasefunc = c->d_bt_phasefunc; ca.pulindex = c->d_bt_pulindex; ca.st = c->d_bst;
    for (int i = 0; i < LRH_BLN_INFO_SIZE; i++) { ca.bln_size[i] = c->bt.bln[i].size; ca.bln_rest[i] = c->bt.bln[i].rest; ca.bln_avgmax[i] = c->bt.bln[i].avgmax; }
    { const int wn = std::max(c->bt.bln[c->bt.largest_blnfit].size / 2, ca.pwid) + 1;
      // A pulse reaches R samples ahead (the search) and wn to either side (the fit); the residue of a subtracted pulse can be a
      // new candidate up to wn away with the same reach again.  Twice that keeps neighbouring extents apart on every signal tried
      // (half of it -- R + 2 wn + 16 -- collided in most calls of the full-size test, and one collision sends the whole span to
      // the one-wave replay).
      ca.gap = std::max(64, 2 * (ca.R + 2 * wn)); }
    ca.bk_margin = 256;
    const size_t need = (size_t)a.total + 2 * ca.bk_margin + 1;
    if (c->clv_cap < need) {
      HIPCHK(c, hipStreamSynchronize(c->cur));
      for (void **q_ : { (void **)&c->d_clv_start, (void **)&c->d_clv_ext, (void **)&c->d_clv_dbg, (void **)&c->d_clv_ctl, (void **)&c->d_clv_bk_pos, (void **)&c->d_clv_bk_pwr, (void **)&c->d_clv_bk_tf, (void **)&c->d_clv_bk_pwo, (void **)&c->d_clv_bk_ty })
        if (*q_) { lrh_dev_free(*q_); *q_ = nullptr; }
      c->clv_cap = 0;
      const size_t cap = need + need / 4;
      const int maxr = (int)(cap / ca.gap) + 2;
      int rc_ = LRH_OK;
      if ((rc_ = dev_alloc(c, &c->d_clv_start, maxr)) || (rc_ = dev_alloc(c, &c->d_clv_ext, 2 * (size_t)maxr)) || (rc_ = dev_alloc(c, &c->d_clv_ctl, 8 + 1024)) || (getenv("LRH_CLEVER_DEBUG") && (rc_ = dev_alloc(c, &c->d_clv_dbg, 2 * (size_t)maxr))) || (rc_ = dev_alloc(c, &c->d_clv_bk_pos, cap, false)) ||
          (!c->d_clv_logged && (rc_ = dev_alloc(c, &c->d_clv_logged, (size_t)c->cfg.timf2pow_size / 64))) ||
          (rc_ = dev_alloc(c, &c->d_clv_bk_pwr, cap, false)) || (rc_ = dev_alloc(c, &c->d_clv_bk_tf, cap, false))) return rc_;
      if (coupled && ((rc_ = dev_alloc(c, &c->d_clv_bk_pwo, cap, false)) || (rc_ = dev_alloc(c, &c->d_clv_bk_ty, cap, false)))) return rc_;
      c->clv_cap = cap; c->clv_max_regions = maxr;
    }
    ca.reg_start = c->d_clv_start; ca.reg_ext = c->d_clv_ext; ca.reg_dbg = c->d_clv_dbg; ca.reg_ctl = c->d_clv_ctl; ca.max_regions = c->clv_max_regions;
    ca.logged = c->d_clv_logged; ca.bk_pos = c->d_clv_bk_pos; ca.bk_pwr = c->d_clv_bk_pwr; ca.bk_tf = c->d_clv_bk_tf; ca.bk_pwo = c->d_clv_bk_pwo; ca.bk_ty = c->d_clv_bk_ty; ca.force_serial = c->clever_force_serial ? 1 : 0;
    if (c->rec) {
      // deferred schedule: the search is parked with the rest of the round's launches; the bookkeeping that depends on where it stops,
      // the statistics and the dumb blanker follow when the next blanker call (or the end of lrh_wideband_dsp) asks for the resume point
      if (!c->h_clv_out) {
        if (lrh_host_malloc((void **)&c->h_clv_out, 4 * sizeof(int)) != hipSuccess) return fail(c, LRH_ENOMEM, "hipHostMalloc");
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_clv, hipEventDisableTiming));
      }
      // liminfo_amplitude_factor as the limiter run BEFORE this round's left it (the serial order: search, then this round's limiter): the
      // in-call limiter of the round has already been enqueued on its own stream, behind a copy of the factor taken for this search
      if (c->wl_on && c->d_clv_amp && c->clv_amp_seq > 0) ca.amp_dev = c->d_clv_amp + ((c->clv_amp_seq - 1) & 1);
      hipEvent_t ev_amp = ca.amp_dev ? c->ev_amp : nullptr;
      // (LRH_CLEVER_SPLIT=1) Candidate bits and the list of regions only read the power ring: they may go out now, on the side stream
      // behind the previous round's dumb blanker (whose update has made the limit final) and this round's make_timf2, so that the
      // replay starts the next round with its list made.  Measured: the replay's stage shrinks (536 -> 374 us) but the bandwidth-bound
      // front then runs beside fft2 of the previous round and costs it more (431 -> 490 us): 25.2 against 26.1 Gsamples/s.  Off.
      int back_parts = 3;
      if (c->clv_split && c->clv_ev_t2) {
        hipStream_t keep_cur = c->cur; std::vector<std::function<int(lrh_ctx *)>> *keep_rec = c->rec;
        c->cur = c->stream2; c->rec = nullptr;
        hipError_t e_ = hipStreamWaitEvent(c->stream2, c->clv_ev_t2, 0);
        if (e_ == hipSuccess) { ProfScope ps(c, "clever"); e_ = launch_clever(ca, c->stream2, 1); }
        c->cur = keep_cur; c->rec = keep_rec;
        if (e_ != hipSuccess) return fail(c, LRH_EDEVICE, "launch_clever (front)", e_);
        back_parts = 2;
      }
      LRH_DEVICE_WORK(c, {
        if (ev_amp) HIPCHK(c, hipStreamWaitEvent(c->cur, ev_amp, 0));
        { ProfScope ps(c, "clever"); HIPCHK(c, launch_clever(ca, c->cur, back_parts)); }
        HIPCHK(c, hipMemcpyAsync(c->h_clv_out, (char *)c->d_bst + offsetof(BlankState, clever_out), 4 * sizeof(int), hipMemcpyDeviceToHost, c->cur));
        HIPCHK(c, hipEventRecord(c->ev_clv, c->cur));
        c->clv_issued = true;
      });
      c->clv_late.a = a; c->clv_late.ca = ca; c->clv_late.pbeg = pbeg; c->clv_late.lowlevel = p->fft1_lowlevel_fraction;
      c->clv_wait = true; c->clv_issued = false;
      return LRH_OK;
    }
    int out[4];
    { ProfScope ps(c, "clever"); HIPCHK(c, launch_clever(ca, c->cur)); }
    HIPCHK(c, stage_d2h(c, out, (char *)c->d_bst + offsetof(BlankState, clever_out), sizeof out, c->cur));
    HIPCHK(c, hipStreamSynchronize(c->cur));
    if (out[3]) {                                           // colliding extents: samples back from the undo log, one wave over the span
      { ProfScope ps(c, "clever"); HIPCHK(c, launch_clever(ca, c->cur, 4)); }
      HIPCHK(c, stage_d2h(c, out, (char *)c->d_bst + offsetof(BlankState, clever_out), sizeof out, c->cur));
      HIPCHK(c, hipStreamSynchronize(c->cur));
    }
    { static const int dbg = getenv("LRH_CLEVER_DEBUG") ? atoi(getenv("LRH_CLEVER_DEBUG")) : 0;     // diagnostics: the regions of this call and where extents met
      if (dbg) {
        int ctl[4] = {0, 0, 0, 0}; stage_d2h(c, ctl, c->d_clv_ctl, sizeof ctl, c->stream);
        const int nr = std::min(ctl[0], c->clv_max_regions);
        std::vector<int> st(nr), ex(2 * (size_t)nr);
        if (nr) { stage_d2h(c, st.data(), c->d_clv_start, nr * sizeof(int), c->stream); stage_d2h(c, ex.data(), c->d_clv_ext, 2 * nr * sizeof(int), c->stream); }
        int bad = 0, first_bad = -1;
        for (int r = 0; r + 1 < nr; r++) if (ex[2 * r + 1] >= ex[2 * (r + 1)]) { if (first_bad < 0) first_bad = r; bad++; }
        fprintf(stderr, "clever: total %d regions %d (max %d) serial %d gap %d colliding pairs %d", a.total, ctl[0], c->clv_max_regions, ctl[1], ca.gap, bad);
        if (first_bad >= 0) fprintf(stderr, "  first: region %d start %d ext [%d, %d] | region %d start %d ext [%d, %d]", first_bad, st[first_bad], ex[2 * first_bad], ex[2 * first_bad + 1],
                                    first_bad + 1, st[first_bad + 1], ex[2 * first_bad + 2], ex[2 * first_bad + 3]);
        fprintf(stderr, "  fitted %d rejected %d\n", out[1], out[2]);
        if (c->d_clv_dbg && nr) {
          std::vector<int> dbg(2 * (size_t)nr); stage_d2h(c, dbg.data(), c->d_clv_dbg, 2 * nr * sizeof(int), c->stream);
          int hist[8] = {0}, worst = 0; long long ticks = 0;
          for (int r = 0; r < nr; r++) { hist[std::min(7, dbg[2 * r] / 2)]++; ticks += dbg[2 * r + 1]; if (dbg[2 * r + 1] > dbg[2 * worst + 1]) worst = r; }
          fprintf(stderr, "clever: candidates per region 0-1 %d, 2-3 %d, 4-5 %d, 6-7 %d, 8-9 %d, 10-11 %d, 12-13 %d, more %d; mean %.1f us per region, slowest region %d: %d candidates, %.1f us, %d samples\n",
                  hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], 0.01 * ticks / nr, worst, dbg[2 * worst], 0.01 * dbg[2 * worst + 1],
                  (worst + 1 < nr ? st[worst + 1] : a.total) - st[worst]);
        }
      } }
    return blanker_tail(c, p, a, pbeg, out, p->fft1_lowlevel_fraction, coupled);
  }
  return blanker_tail(c, p, a, pbeg, nullptr, p->fft1_lowlevel_fraction, coupled);
}
LRH_CATCH(c)

// ---- two coupled RF channels: see include/linrad_hip.h
int lrh_blanker_begin(lrh_ctx *c, const lrh_ptrs *p, int *count)
try {
  LRH_ENTER(c);
  if (!c || !p || !count) return LRH_EINVAL;
  if (c->cfg.blanker_channels != 2) return fail(c, LRH_ESTATE, "blanker_channels != 2");
  const int mask = c->timf2pow_mask, pbeg = p->timf2p_fit;
  int pend = (p->timf2_pa / 4 - c->cfg.blnfit_range + mask) & mask;
  pend &= 0xfffffffc;
  *count = 0; c->x_count = -1;
  if (((pend - pbeg + 1 + mask) & mask) < c->cfg.blanker_min_points) return LRH_OK;
  c->x_pbeg = pbeg; c->x_span = (pend - pbeg) & mask; c->xw_count = 0;
  // linear blanker: its search and fits read (and rewrite) up to blnfit_range samples beyond the span, and both channels' samples that far to either side
  const int R = c->clever_on ? c->cfg.blnfit_range : 0;
  if (c->clever_on && c->x_span > c->cfg.timf2pow_size - 1024) { c->x_count = -1; return fail(c, LRH_EINVAL, "linear blanker: span longer than the timf2 power ring"); }
  c->x_count = c->x_span + R;
  { ProfScope ps(c, "xcopy"); HIPCHK(c, launch_span_copy(c->d_xbuf, c->d_pwr, pbeg, c->x_count, mask, 0, c->cur)); }
  if (c->clever_on) {
    const int nw = c->x_span + 2 * R + 1;
    HIPCHK(c, launch_span_copy2(c->d_xweak + (size_t)(c->cfg.timf1_channel_index & 1) * nw, c->d_timf2w, pbeg - R, nw, mask, 0, c->cur));
    c->xw_count = 2 * nw;
  }
  *count = c->x_count;
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_blanker_weak_span(lrh_ctx *c, size_t *count)
try {
  LRH_ENTER(c);
  if (!c || !count) return LRH_EINVAL;
  if (c->cfg.blanker_channels != 2) return fail(c, LRH_ESTATE, "blanker_channels != 2");
  *count = c->x_count > 0 ? (size_t)c->xw_count : 0;
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_blanker_finish(lrh_ctx *c, lrh_ptrs *p)
try {
  LRH_ENTER(c);
  if (!c || !p) return LRH_EINVAL;
  LRH_WRITES(c, RB(LRH_RING_TIMF2_FLOAT) | RB(LRH_RING_TIMF2_PWR));
  if (c->cfg.blanker_channels != 2 || !c->fin_pending) return fail(c, LRH_ESTATE, "no coupled blanker call to finish");
  c->fin_pending = false;
  HIPCHK(c, launch_blanker(c->fin_args, c->cfg.timf2pow_size / 32, c->cur));
  return LRH_OK;
}
LRH_CATCH(c)
static int exchange_span(lrh_ctx *c, int which, float **ptr, size_t *cap)
{
  if (which == LRH_X_POL) { if (!c->d_xpol) return fail(c, LRH_ESTATE, "fft3 not configured"); *ptr = (float *)c->d_xpol; *cap = (size_t)4 * c->cfg.max_fft3n * c->Nm2; return LRH_OK; }
  if (c->cfg.blanker_channels != 2) return fail(c, LRH_ESTATE, "blanker_channels != 2");
  if (which == LRH_X_WEAK) { if (!c->d_xweak) return fail(c, LRH_ESTATE, "linear blanker tables not installed"); *ptr = (float *)c->d_xweak; *cap = (size_t)4 * c->cfg.timf2pow_size; return LRH_OK; }
  if (which == LRH_X_PWR) { *ptr = c->d_xbuf; *cap = (size_t)c->cfg.timf2pow_size; }
  else if (which == LRH_X_STAT) { *ptr = c->d_xstat; *cap = 2; }
  else if (which == LRH_X_BINS) { *ptr = (float *)c->d_xbins; *cap = (size_t)4 * c->cfg.max_fft2n * c->N2; }
  else if (which == LRH_X_SPEC) { if (!c->d_xspec) return fail(c, LRH_ESTATE, "lrh_set_correlation first"); *ptr = (float *)c->d_xspec; *cap = (size_t)4 * c->cfg.max_batch * c->N1; }
  else if (which == LRH_X_POL) { if (!c->d_xpol) return fail(c, LRH_ESTATE, "fft3 not configured"); *ptr = (float *)c->d_xpol; *cap = (size_t)4 * c->cfg.max_fft3n * c->Nm2; }
  else return LRH_EINVAL;
  return LRH_OK;
}
int lrh_exchange_ptr(lrh_ctx *c, int which, void **device_ptr)
try {
  LRH_ENTER(c);
  if (!c || !device_ptr) return LRH_EINVAL;
  float *q; size_t cap; const int rc = exchange_span(c, which, &q, &cap); if (rc) return rc;
  *device_ptr = q; return LRH_OK;
}
LRH_CATCH(c)
int lrh_exchange_read(lrh_ctx *c, int which, float *dst, size_t off, size_t count)
try {
  LRH_ENTER(c);
  if (!c || !dst) return LRH_EINVAL;
  float *q; size_t cap; const int rc = exchange_span(c, which, &q, &cap); if (rc) return rc;
  if (off + count > cap) return LRH_EINVAL;
  HIPCHK(c, stage_d2h(c, dst, q + off, 4 * count, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_exchange_write(lrh_ctx *c, int which, const float *src, size_t off, size_t count)
try {
  LRH_ENTER(c);
  if (!c || !src) return LRH_EINVAL;
  float *q; size_t cap; const int rc = exchange_span(c, which, &q, &cap); if (rc) return rc;
  if (off + count > cap) return LRH_EINVAL;
  HIPCHK(c, stage_h2d(c, q + off, src, 4 * count, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_get_blanker_state(lrh_ctx *c, lrh_blanker_state *st)
try {
  LRH_ENTER(c);
  if (!c || !st) return LRH_EINVAL;
  BlankState bs;
  HIPCHK(c, stage_d2h(c, &bs, c->d_bst, sizeof bs, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  st->timf2_noise_floor = bs.noise_floor; st->stupid_bln_limit = bs.limit;
  st->timf2_despiked_pwr[0] = bs.despiked_pwr[0]; st->timf2_despiked_pwr[1] = bs.despiked_pwr[1];
  st->timf2_despiked_pwrinc[0] = bs.despiked_pwrinc[0]; st->timf2_despiked_pwrinc[1] = bs.despiked_pwrinc[1];
  st->stupid_blanker_rate = bs.stupid_rate; st->timf2_cleared_points = bs.cleared_acc;
  st->last_call_cleared = bs.last_cleared; st->slow_path_calls = bs.slow_calls;
  st->clever_bln_limit = bs.clever_limit; st->clever_blanker_rate = bs.clever_rate; st->timf2_fitted_pulses = bs.fitted_acc;
  st->last_call_fitted = bs.last_fitted; st->last_call_rejected = bs.last_rejected; st->clever_serial_calls = bs.clever_serial_calls;
  return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- fft2
int lrh_make_fft2(lrh_ctx *c, lrh_ptrs *p, int batch)
try {
  LRH_ENTER(c);
  if (!c || !p || batch < 1 || batch > c->cfg.max_fft2n) return LRH_EINVAL;
  LRH_WRITES(c, RB(LRH_RING_FFT2_FLOAT) | RB(LRH_RING_FFT2_POWER) | RB(LRH_RING_FFT2_POWERSUM) | RB(LRH_RING_WG_WATERF) | RB(LRH_RING_FFT2_XYPOWER) | RB(LRH_RING_FFT2_XYSUM));
  const int N = c->N2;
  Fft2Args a;
  a.timf2w = c->d_timf2w; a.timf2s = c->d_timf2s; a.mask = c->timf2pow_mask; a.px_first = p->timf2_px / 4; a.step = c->M2;
  a.window = c->d_window2; a.tw = c->d_tw2; a.out = c->d_fft2; a.power = c->d_power2; a.first_na = p->fft2_na; a.na_mask = c->fft2n_mask;
  a.xcd = (c->xcd_mask >> 2) & 1; a.run = c->env_fft2_run;
  a.ps_in = c->d_powersum2; a.ps_out = c->d_powersum2_alt; a.wf_scratch = c->d_wf_scratch;
  a.ps_counter = p->wg_waterf_sum_counter; a.ps_avgnum = c->fft2_fused ? c->cfg.waterfall_avgnum : 0;
  Fft2BigArgs g;
  g.timf2w = a.timf2w; g.timf2s = a.timf2s; g.mask = a.mask; g.px_first = a.px_first; g.step = a.step; g.window = a.window;
  g.tw_a = c->d_tw2a; g.tw_b = c->d_tw2b; g.tw_big = c->d_tw2; g.scratch = c->d_fft2_scratch;
  g.out = a.out; g.power = a.power; g.first_na = a.first_na; g.na_mask = a.na_mask;
  g.ps_in = a.ps_in; g.ps_out = a.ps_out; g.wf_scratch = a.wf_scratch; g.ps_counter = a.ps_counter; g.ps_avgnum = a.ps_avgnum; g.batch = batch; g.run = c->env_fft2_cols_run;
  // the band of every transform that reaches the ring: all of it, or what fft2_mix1_fixed will cut out (cfg.fft2_float_sparse)
  a.keep_lo = 0; a.keep_hi = N;
  if (c->cfg.fft2_float_sparse && c->fft2_fused && c->spur_n == 0 && c->cfg.blanker_channels != 2 && c->ms.mix1_selfreq >= 0) {
    const int centre = (int)((float)c->ms.mix1_selfreq * c->cfg.fftx_points_per_hz + 0.5);
    a.keep_lo = std::max(0, centre - c->Nm / 2 - 64); a.keep_hi = std::min(N, centre + c->Nm / 2 + 64);
  }
  g.keep_lo = a.keep_lo; g.keep_hi = a.keep_hi;
  if (c->cfg.fft2_float_sparse) {
    if (c->fft2_keep_lo.empty()) { c->fft2_keep_lo.assign(c->cfg.max_fft2n, 0); c->fft2_keep_hi.assign(c->cfg.max_fft2n, N); }
    for (int b = 0; b < batch; b++) { c->fft2_keep_lo[(p->fft2_na + b) & c->fft2n_mask] = a.keep_lo; c->fft2_keep_hi[(p->fft2_na + b) & c->fft2n_mask] = a.keep_hi; }
  }
  Powersum2Args s;
  s.power = c->d_power2; s.na_mask = c->fft2n_mask; s.first_na = p->fft2_na; s.count = batch; s.n = N;
  s.powersum_in = c->d_powersum2; s.powersum_out = c->d_powersum2_alt; s.wf_scratch = c->d_wf_scratch; s.counter = p->wg_waterf_sum_counter; s.avgnum = c->cfg.waterfall_avgnum;
  { float *t = c->d_powersum2; c->d_powersum2 = c->d_powersum2_alt; c->d_powersum2_alt = t; }   // ping-pong: group 0 reads while the last group writes
  // two coupled channels: the line comes from both channels' sums (lrh_fft2_xy_finish); the pointers advance as usual
  const int nlines = c->cfg.blanker_channels == 2 ? 0 : (p->wg_waterf_sum_counter + batch) / c->cfg.waterfall_avgnum;
  WaterfallArgs w; memset(&w, 0, sizeof w);
  if (nlines > 0) {
    int hx, hp, wx, wp, wfirst; wf_geometry(c, &hx, &hp, &wx, &wp, &wfirst);
    w.ps = c->d_wf_scratch; w.yfac = c->d_yfac; w.itab = c->d_wf_itab; w.line = c->d_waterf;
    w.npix = c->cfg.wf_xpixels; w.first = c->cfg.wf_first_xpoint; w.siz = N; w.hx = hx; w.hp = hp;
    w.ptr0 = p->wg_waterf_ptr; w.wf_size = c->cfg.wf_lines * c->cfg.wf_xpixels; w.line_stride = N;
  }
  const int fft2_n = c->cfg.fft2_n;
  // spurs tracked: eliminate_spurs sits between the transform and the power sums (FFT2_ELIMINATE_SPURS, fft2.c:647-652), so the sums
  // cannot ride inside the transform kernel: transform, k_spur over the batch in order, |X|^2 of the cleaned bins, then the sums
  // spurs tracked: eliminate_spurs sits between the transform and the power sums (fft2.c:647-652).  The fused form stays: the transform
  // kernel sums |X|^2 as always, k_spur takes the carriers out of the batch in order, and k_spur_patch redoes the sums of the few bins
  // it touched; without the fused form: transform, k_spur, |X|^2 of the cleaned bins, k_powersum2
  const bool spurs = c->spur_n > 0;
  const bool fused = c->fft2_fused;
  SpurArgs sa; memset(&sa, 0, sizeof sa);
  if (spurs) spur_args(c, &sa, p->fft2_na, batch);
  const int na0 = p->fft2_na, max2 = c->cfg.max_fft2n;
  SpurPatchArgs pa; memset(&pa, 0, sizeof pa);
  pa.fft2 = c->d_fft2; pa.n = N; pa.first_na = p->fft2_na; pa.na_mask = c->fft2n_mask; pa.count = batch; pa.counter = s.counter; pa.avgnum = s.avgnum;
  pa.touched = c->d_spur_touched; pa.powersum_in = s.powersum_in; pa.powersum_out = s.powersum_out; pa.wf_scratch = s.wf_scratch;
  LRH_DEVICE_WORK(c, {
    if (c->split_fft2_tail) HIPCHK(c, hipStreamWaitEvent(c->cur, c->ev_ps2, 0));   // side-stream sums of the previous call read these rings
    if (fft2_n <= 14) { ProfScope ps(c, "fft2"); HIPCHK(c, launch_fft2(fft2_n, a, batch, c->cur)); }
    else if (c->fft2_span > 0 && g.ps_avgnum > 0 && batch >= 2 * c->fft2_span && 5 * (c->fft2_span + g.ps_avgnum) <= c->cfg.max_fft2n) {
      ProfScope ps(c, "fft2");
      if (!c->stream_f2) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream_f2, hipStreamNonBlocking));
        for (int i = 0; i < 3; i++) { HIPCHK(c, hipEventCreateWithFlags(&c->ev_f2c[i], hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_f2r[i], hipEventDisableTiming)); }
      }
      // spans end where a waterfall averaging group ends (the row step sums |X|^2 over whole groups): the first carries the group in
      // progress, the last may leave one unfinished
      const int avg = g.ps_avgnum;
      const int span = (c->fft2_span + avg - 1) / avg * avg;
      { static int said = 0; if (!said++ && getenv("LRH_FFT2_SPAN_DEBUG")) fprintf(stderr, "fft2 of %d transforms in spans of %d\n", batch, span); }
      int b0 = 0;
      int sp = 0;
      while (b0 < batch) {
        int b1 = b0 + span - (b0 == 0 ? g.ps_counter % avg : 0);          // (ps_counter + b1) % avg == 0
        if (batch - b1 < span / 2) b1 = batch;
        Fft2BigArgs gs = g;
        const int slot = sp % 3;
        gs.px_first = (g.px_first + b0 * g.step) & g.mask; gs.first_na = (g.first_na + b0) & g.na_mask;
        gs.scratch = g.scratch + (size_t)slot * (span + span / 2) * N;
        gs.ps_counter = b0 == 0 ? g.ps_counter : 0;
        gs.wf_scratch = g.wf_scratch + (size_t)((g.ps_counter + b0) / avg) * N;
        if (sp >= 3) HIPCHK(c, hipStreamWaitEvent(c->cur, c->ev_f2r[slot], 0));          // the row step that read this scratch slot last
        HIPCHK(c, launch_fft2_big(fft2_n, gs, b1 - b0, c->cur, 1));
        HIPCHK(c, hipEventRecord(c->ev_f2c[slot], c->cur));
        HIPCHK(c, hipStreamWaitEvent(c->stream_f2, c->ev_f2c[slot], 0));
        HIPCHK(c, launch_fft2_big(fft2_n, gs, b1 - b0, c->stream_f2, 2));
        HIPCHK(c, hipEventRecord(c->ev_f2r[slot], c->stream_f2));
        b0 = b1; sp++;
      }
      for (int i = 0; i < 3 && i < sp; i++) HIPCHK(c, hipStreamWaitEvent(c->cur, c->ev_f2r[(sp - 1 - i) % 3], 0));
    }
    else { ProfScope ps(c, "fft2"); HIPCHK(c, launch_fft2_big(fft2_n, g, batch, c->cur)); }
    hipStream_t main_s = c->cur;
    if (c->split_fft2_tail) {                      // power sums and waterfall lines only feed the GUI side: side stream
      HIPCHK(c, hipEventRecord(c->ev_fft2, main_s)); HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fft2, 0));
      c->cur = c->stream2;
    }
    if (spurs) {
      ProfScope ps(c, "spur");
      HIPCHK(c, launch_spur(sa, main_s));
      if (fused) HIPCHK(c, launch_spur_patch(pa, sa.nspurs, (s.counter + batch + s.avgnum - 1) / s.avgnum, main_s));
      else {
        const int first = batch < max2 - na0 ? batch : max2 - na0;          // the batch's ring slots may wrap once
        HIPCHK(c, launch_power_of(c->d_fft2 + (size_t)na0 * N, c->d_power2 + (size_t)na0 * N, (size_t)first * N, main_s));
        if (batch > first) HIPCHK(c, launch_power_of(c->d_fft2, c->d_power2, (size_t)(batch - first) * N, main_s));
      }
      if (c->split_fft2_tail) { HIPCHK(c, hipEventRecord(c->ev_fft2, main_s)); HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fft2, 0)); }
    }
    if (c->d_ss_sum) {                                     // the search for new spurs takes the batch's power rows one by one (fft2.c:673-699)
      for (int b = 0; b < batch; b++) { const int rcs_ = spur_search_row(c, (na0 + b) & c->fft2n_mask, main_s); if (rcs_) return rcs_; }
    }
    if (!fused) { ProfScope ps(c, "powersum2"); HIPCHK(c, launch_powersum2(s, c->cur)); }
    if (nlines > 0) { ProfScope ps(c, "waterfall"); HIPCHK(c, launch_waterfall(w, nlines, c->cur)); }
    if (c->split_fft2_tail) { HIPCHK(c, hipEventRecord(c->ev_ps2, c->stream2)); c->last_main_ev = c->ev_fft2; }
    c->cur = main_s;
    { const int rcm_ = stage_mark(c, LRH_STAGE_FFT2); if (rcm_) return rcm_; }
  });
  for (int b = 0; b < batch; b++) {                                      // fft2.c:672, 703-705, 813-815, 1831-1845
    p->wg_waterf_sum_counter++;
    if (p->wg_waterf_sum_counter >= c->cfg.waterfall_avgnum) {
      p->wg_waterf_ptr -= c->cfg.wf_xpixels; if (p->wg_waterf_ptr < 0) p->wg_waterf_ptr += c->cfg.wf_lines * c->cfg.wf_xpixels;
      p->wg_waterf_sum_counter = 0; p->fft2_liminfo_cnt++;
    }
    p->timf2_px = (p->timf2_px + 4 * c->M2) & c->timf2_mask;
    p->fft2_na = (p->fft2_na + 1) & c->fft2n_mask; p->fft2_pa = 2 * p->fft2_na * N;
    p->fft2_nb = (p->fft2_nb + 1) & c->fft2n_mask;
    if (p->fft2_nm != c->fft2n_mask) p->fft2_nm++;
  }
  return LRH_OK;
}
LRH_CATCH(c)

// Two coupled channels (include/linrad_hip.h): the new transforms of the own channel go to their slot of LRH_X_BINS ...
int lrh_fft2_xy_begin(lrh_ctx *c, const lrh_ptrs *at, int batch, size_t *count)
try {
  LRH_ENTER(c);
  if (!c || !at || !count || batch < 1 || batch > c->cfg.max_fft2n) return LRH_EINVAL;
  if (c->cfg.blanker_channels != 2) return fail(c, LRH_ESTATE, "blanker_channels != 2");
  const int N = c->N2, na = at->fft2_na & c->fft2n_mask;
  float2 *slot = c->d_xbins + (size_t)(c->cfg.timf1_channel_index & 1) * batch * N;
  const int first = std::min(batch, c->cfg.max_fft2n - na);          // the ring span may wrap once
  LRH_DEVICE_WORK(c, {
    ProfScope ps(c, "xcopy");
    HIPCHK(c, hipMemcpyAsync(slot, c->d_fft2 + (size_t)na * N, (size_t)first * N * sizeof(float2), hipMemcpyDeviceToDevice, c->cur));
    if (batch > first) HIPCHK(c, hipMemcpyAsync(slot + (size_t)first * N, c->d_fft2, (size_t)(batch - first) * N * sizeof(float2), hipMemcpyDeviceToDevice, c->cur));
  });
  *count = (size_t)batch * 2 * N;
  return LRH_OK;
}
LRH_CATCH(c)
// ... and with the partner's slot in place: TWOCHAN_POWER per transform, fft2_xysum, and the waterfall lines that complete
// within the batch (fft2.c:1622-1640, 1700-1815)
int lrh_fft2_xy_finish(lrh_ctx *c, const lrh_ptrs *at, int batch)
try {
  LRH_ENTER(c);
  if (!c || !at || batch < 1 || batch > c->cfg.max_fft2n) return LRH_EINVAL;
  LRH_WRITES(c, RB(LRH_RING_FFT2_XYPOWER) | RB(LRH_RING_FFT2_XYSUM) | RB(LRH_RING_WG_WATERF));
  if (c->cfg.blanker_channels != 2) return fail(c, LRH_ESTATE, "blanker_channels != 2");
  const int N = c->N2;
  XyArgs a;
  a.x = c->d_xbins; a.y = c->d_xbins + (size_t)batch * N;
  if (c->xy_own_src) { if (c->cfg.timf1_channel_index & 1) a.y = c->xy_own_src; else a.x = c->xy_own_src; }   // lrh_wideband_dsp: the own channel straight from the ring
  a.xypower = c->cfg.fft2_float_sparse ? nullptr : c->d_xypower; a.first_na = at->fft2_na; a.na_mask = c->fft2n_mask;
  a.n = N; a.batch = batch; a.sum_in = c->d_xysum; a.sum_out = c->d_xysum_alt; a.lines = c->d_wf_scratch;
  a.counter = at->wg_waterf_sum_counter; a.avgnum = c->cfg.waterfall_avgnum;
  { float4 *t = c->d_xysum; c->d_xysum = c->d_xysum_alt; c->d_xysum_alt = t; }   // ping-pong: group 0 reads while the last group writes
  const int nlines = (at->wg_waterf_sum_counter + batch) / c->cfg.waterfall_avgnum;
  WaterfallArgs w; memset(&w, 0, sizeof w);
  if (nlines > 0) {
    int hx, hp, wx, wp, wfirst; wf_geometry(c, &hx, &hp, &wx, &wp, &wfirst);
    w.ps = c->d_wf_scratch; w.yfac = c->d_yfac; w.itab = c->d_wf_itab; w.line = c->d_waterf;
    w.npix = c->cfg.wf_xpixels; w.first = c->cfg.wf_first_xpoint; w.siz = N; w.hx = hx; w.hp = hp;
    w.ptr0 = at->wg_waterf_ptr; w.wf_size = c->cfg.wf_lines * c->cfg.wf_xpixels; w.line_stride = N;
  }
  LRH_DEVICE_WORK(c, {
    if (c->split_fft2_tail) HIPCHK(c, hipStreamWaitEvent(c->cur, c->ev_ps2, 0));   // side-stream sums of make_fft2 share wf_scratch
    { ProfScope ps(c, "xypower"); HIPCHK(c, launch_xypower(a, c->cur)); }
    if (nlines > 0) { ProfScope ps(c, "waterfall"); HIPCHK(c, launch_waterfall(w, nlines, c->cur)); }
  });
  return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- mix1
int lrh_set_mix1_selfreq(lrh_ctx *c, double fq) try { LRH_LOCK(c); if (!c) return LRH_EINVAL; c->ms.mix1_selfreq = fq; return LRH_OK; } LRH_CATCH(c)
int lrh_get_mix1_state(lrh_ctx *c, lrh_mix1_state *st) try { LRH_ENTER(c); if (!c || !st) return LRH_EINVAL; *st = c->ms; return LRH_OK; } LRH_CATCH(c)

// Tuning of one mix1 transform (what set_mix1_phases, mix1.c:781-861, decides; float branch -- the double branch belongs to
// correlation mode).  The selected frequency splits into the fft bin the baseband block is cut around, the block-to-block phase
// advance of that bin, and a per-sample rotation for the fraction of a bin that is left.  The operations and their float
// roundings are the reference's: the phases feed serial recursions whose rounding shows in timf3 at the 1e-5 level.
static int mix1_retune(lrh_ctx *c, float hz)
{
  lrh_mix1_state &m = c->ms;
  if (hz < c->cfg.mix1_lowest_fq || hz > c->cfg.mix1_highest_fq) return LRH_ERANGE;
  const int block_bins = c->Nm;
  const float in_bins = hz * c->cfg.fftx_points_per_hz;
  const int centre_bin = (int)(in_bins + 0.5);
  const int bin_in_block = centre_bin % block_bins;
  const float whole_blocks = (float)(block_bins * (centre_bin / block_bins));
  float fraction = in_bins - whole_blocks - bin_in_block;
  fraction = fraction - (int)(fraction);
  m.mix1_phase_rot = (float)(fraction * 2 * PI_L / block_bins);
  const int advance_bins = (bin_in_block * c->Mm) % block_bins;     // phase of that bin after the Mm new samples of a block
  m.mix1_old_phase = m.mix1_phase;
  m.mix1_phase += m.mix1_phase_step;
  m.mix1_phase_step = (float)(advance_bins * 2 * PI_L / block_bins);
  m.mix1_old_point = (m.mix1_point != -1) ? m.mix1_point : centre_bin;
  m.mix1_point = centre_bin;
  // the reference folds with "< pi" on the second test, so the phase always ends up raised by 2 pi (mix1.c:859-860); kept
  if (m.mix1_phase > PI_L) m.mix1_phase = (float)(m.mix1_phase - 2 * PI_L);
  if (m.mix1_phase < PI_L) m.mix1_phase = (float)(m.mix1_phase + 2 * PI_L);
  return LRH_OK;
}

// Bookkeeping of the AFC's per-transform frequency tables that do_mix1_afc (mix1.c:648-768) does before it calls do_mix1: the
// drift handed to the next transform may not change faster than a fraction of the baseband bandwidth per transform.  When the
// supplied track asks for more, the tables ahead are rewritten: accelerate at the limit until half the miss is made up, then
// brake at the limit for as many transforms.  `ring` wraps the caller's four tables (the reference's globals of the same names).
#define LRH_BWFAC 0.03
namespace {
struct AfcRing {
  float *mid, *slope, *curv, *start; int mask;
  int next(int i) const { return (i + 1) & mask; }
  int prev(int i) const { return (i + mask) & mask; }
};
}
static void afc_tables(lrh_ctx *c, lrh_afc *afc, int now, int newest, int mask)
{
  const AfcRing r{afc->mix1_fq_mid, afc->mix1_fq_slope, afc->mix1_fq_curv, afc->mix1_fq_start, mask};
  const int before = r.prev(now), after = r.next(now);
  const float max_curv = (float)(LRH_BWFAC * afc->baseband_bw_hz);
  float predicted = r.mid[now] + r.slope[before];
  if (fabs(r.mid[after] - predicted) < LRH_BWFAC * afc->baseband_bw_hz) {
    // the track is smooth enough: plain first and second differences
    r.slope[now] = r.mid[after] - r.mid[now];
    r.curv[now] = r.slope[now] - r.slope[before];
  } else {
    float miss = r.mid[after] - predicted;
    float bend = miss < 0 ? -max_curv : max_curv;
    const float half_miss = (float)(fabs(miss) / 2);
    int at = now, from = before, to = after, steps = 0;
    while (fabs(miss) > half_miss && at != newest) {           // accelerate, keeping the rewritten track inside the mix1 range
      r.curv[at] = bend; r.slope[at] = r.slope[from] + bend;
      predicted = r.mid[at] + r.slope[at];
      miss = r.mid[to] - predicted;
      if (predicted < c->cfg.mix1_lowest_fq) predicted = c->cfg.mix1_lowest_fq;
      if (predicted > c->cfg.mix1_highest_fq) predicted = c->cfg.mix1_highest_fq;
      r.mid[to] = predicted;
      from = r.next(from); at = r.next(at); to = r.next(to); steps++;
    }
    const float miss_at_turn = miss;
    bend = -bend;
    while (steps > 0 && at != newest && miss_at_turn * miss > 0) {   // brake for as many transforms, or until the miss changes sign
      r.curv[at] = bend; r.slope[at] = r.slope[from] + bend;
      predicted = r.mid[at] + r.slope[at];
      miss = r.mid[to] - predicted;
      r.mid[to] = predicted;
      from = r.next(from); at = r.next(at); to = r.next(to); steps--;
    }
  }
  r.start[after] = (float)(r.mid[now] + 0.5 * r.slope[now] + 0.25 * r.curv[now]);
}

// afc != nullptr: per-transform frequency from afc->mix1_fq_mid[nx] with the table bookkeeping after each transform;
// nx0 / na / ring_mask describe the source ring position of the first transform (fft2_nx or fft1_nx).
static int mix1_run(lrh_ctx *c, lrh_ptrs *p, int batch, const float2 *src, int n2, int first, int mask, int lim_hi,
                    lrh_afc *afc = nullptr, int na = 0)
{
  LRH_WRITES(c, RB(LRH_RING_TIMF3_FLOAT));
  if (src == c->d_fft1) { const int rc_ = join_handles(c); if (rc_) return rc_; }     // second fft off: the fft1 workers' (or a parked) transforms
  const int Nm = c->Nm, overlap = c->Im != 0, half = c->Mm, block2 = c->Mm;     // block in complex samples = rotated samples per transform
  lrh_mix1_state *s = &c->ms;
  const int selected = s->mix1_selfreq >= 0;
  Mix1OutArgs o; memset(&o, 0, sizeof o);
  o.timf3 = c->d_timf3; o.mask2 = c->cfg.timf3_size / 2 - 1; o.pa_first = p->timf3_pa / 2; o.block = block2;
  o.nm = Nm; o.overlap = overlap; o.selected = selected; o.scratch = c->d_mix_scratch; o.rotate = 1;
  o.xover = (overlap && c->Im != c->Mm) ? c->Xm : 0; o.im = c->Im; o.win = c->d_mixwin; o.sin2win = c->d_sin2win; o.cos2win = c->d_cos2win;
  if (overlap && c->Im != c->Mm && c->Xm < 1) return fail(c, LRH_EINVAL, "mix1 window without a crossover region");
  if (selected) {
    // phase recursions of do_mix1 in the reference's float arithmetic (mix1.c:143-154, 164-187); serial by nature, tiny
    const int slot = c->ph_next; c->ph_next = (c->ph_next + 1) % LRH_NSTAGE;
    if (c->ph_pending[slot]) return fail(c, LRH_ESTATE, "mix1 staging ring exhausted by deferred work");
    { const auto w0 = std::chrono::steady_clock::now();
      timespec tc0; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &tc0);
      struct WaitCpu { lrh_ctx *c; timespec t0; ~WaitCpu() { timespec t1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1); c->host_cpu_ms_wait += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6; } } wait_cpu{c, tc0};
      // the host runs up to LRH_NSTAGE rounds ahead and then waits here for most of a round: asleep, not spinning inside
      // hipEventSynchronize (which burnt a whole core: thread CPU time = wall time of the call)
      // (small rounds finish within tens of microseconds: poll that long first, a sleep would cost the round its own length)
      // ... and after a long wait the next one is long too (the host is LRH_NSTAGE rounds ahead of rounds of ~1 ms): no polling phase then,
      // and sleeps of a fifth of the last wait -- 60 us of spinning and a wake-up every 40 us were 1.2 of the 3.7 ms of CPU time per call of 8 rounds
      const double last = c->ph_last_wait_us;
      long nap_ns = last > 400.0 ? (long)(last * 200.0) : 40000; if (nap_ns > 250000) nap_ns = 250000;
      static const bool block_env = getenv("LRH_STAGE_BLOCK") && atoi(getenv("LRH_STAGE_BLOCK"));   // (measured: the runtime spins inside hipEventSynchronize all the same, 5.4 against 3.3 ms of CPU per call: off)
      for (;;) {
        const hipError_t q = hipEventQuery(c->ph_ev[slot]);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(c, LRH_EDEVICE, "hipEventQuery(staging)", q);
        // a long wait is coming (the last one was): one blocking wait on the event (created with hipEventBlockingSync: the thread sleeps on the
        // interrupt) instead of a dozen timed naps, each a system call and a wake-up -- 0.8 of the 3.3 ms of CPU time per call of 8 rounds
        if (block_env && last > 400.0) { const hipError_t e = hipEventSynchronize(c->ph_ev[slot]); if (e != hipSuccess) return fail(c, LRH_EDEVICE, "hipEventSynchronize(staging)", e); break; }
        const double waited = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        if (last <= 400.0 && waited < 60.0) continue;
        // a long wait (the last one was): one sleep to 85 % of it, then short naps -- every nap is a system call and a wake-up, 5-8 us of CPU each
        if (last > 400.0) { const double left = 0.85 * last - waited; nap_ns = left > 30.0 ? (long)(left * 1000.0) : 30000; if (nap_ns > 2000000) nap_ns = 2000000; }
        timespec ts{0, nap_ns}; nanosleep(&ts, nullptr);
      }
      c->ph_last_wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
      c->host_ms_wait += c->ph_last_wait_us * 1e-3; }
    const int nchunks = (half + LRH_PH_CHUNK - 1) / LRH_PH_CHUNK;
    // the page-locked table of this round: per transform the increments (t2, r2), the phases at its first sample, the bin (k_phase_expand reads it)
    float2 *h_inc = (float2 *)(c->h_ph + slot * c->ph_stride), *h_st = h_inc + batch;
    int point = 0;
    const auto host_t0 = std::chrono::steady_clock::now();
    int *h_point = (int *)(h_st + batch);
    // the reference tests the range before it touches any state (mix1.c:787-796): every frequency of the batch first;
    // a table entry that only becomes invalid through the AFC bookkeeping of an earlier transform of the same batch still
    // ends the call, with the phase state put back (the caller's tables keep what the earlier transforms wrote, as after
    // that many single calls of the reference)
    for (int b = 0; b < batch; b++) {
      const float fq = afc ? afc->mix1_fq_mid[(first + b) & mask] : (float)s->mix1_selfreq;
      if (fq < c->cfg.mix1_lowest_fq || fq > c->cfg.mix1_highest_fq) { c->ph_next = slot; return LRH_ERANGE; }
    }
    const lrh_mix1_state ms_keep = *s;
    for (int b = 0; b < batch; b++) {
      const int nx = (first + b) & mask;
      int rc = mix1_retune(c, afc ? afc->mix1_fq_mid[nx] : (float)s->mix1_selfreq);
      if (rc) { *s = ms_keep; c->ph_next = slot; return rc; }
      if (afc) afc_tables(c, afc, nx, na, mask);
      point = s->mix1_point; h_point[b] = point;
      if (src == c->d_fft2 && !c->fft2_keep_lo.empty()) {              // cfg.fft2_float_sparse: the band must be what lrh_make_fft2 kept of this transform
        const int lo = std::max(0, point - Nm / 2), hi = std::min(lim_hi, point + Nm / 2);
        if (lo < c->fft2_keep_lo[nx] || hi > c->fft2_keep_hi[nx]) { *s = ms_keep; c->ph_next = slot;
          return fail(c, LRH_ESTATE, "fft2_float_sparse: the selected frequency has moved off the band stored for this transform"); }
      }
      const float t2 = s->mix1_phase_rot, t1 = s->mix1_phase;
      const float r1 = s->mix1_old_phase;
      const float r2 = overlap ? (float)(t2 - 2 * (s->mix1_old_point - s->mix1_point) * PI_L / Nm) : 0.f;
      h_inc[b] = make_float2(t2, r2);
      h_st[b] = make_float2(t1, r1);
      // do_mix1 adds t2 to the phase once per output sample, in float (mix1.c:141-195): `half` additions, advanced in closed form, bit for bit
      // (lrh_phase.h; the loop `r1 += r2; t1 += t2` cost 0.23 ms per round of 1024 transforms -- more than half of the call's host time);
      // the phases at the chunk starts inside the transform, which the device replays from, are derived on the device by the same function
      s->mix1_phase = lrh_phase_advance(t1, t2, half);
    }
    c->host_ms_phases += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count(); c->host_n_phases++;
    float2 *d_inc = (float2 *)(c->d_ph + slot * c->ph_stride);
    o.ph_inc = d_inc; o.ph_start = d_inc + batch; o.nchunks = nchunks;
    Mix1Args a;
    a.fft2 = src; a.n2 = n2; a.first_nx = first; a.nx_mask = mask; a.fqwin = c->d_fqwin; a.tw = c->d_twm;
    a.scratch = c->d_mix_scratch; a.point = point; a.nm = Nm; a.lim_hi = lim_hi;
    int *const d_point = (int *)(d_inc + (size_t)batch * (1 + nchunks));
    a.points = afc ? (const int *)d_point : nullptr;
    const int mix1_n = c->mix1_n;
    // the device's view of the page-locked table
    const char *const hdev = (const char *)c->h_ph_dev + ((const char *)h_inc - (const char *)c->h_ph);
    const float2 *const g_inc = (const float2 *)hdev, *const g_st = g_inc + batch; const int *const g_point = (const int *)(g_st + batch);
    float2 *const d_start = d_inc + batch;
    if (!c->h_ph_dev) return fail(c, LRH_EDEVICE, "no device view of the page-locked phase table");
    if (c->rec && c->early_upload) {
      // parked kernels: the table goes up now on the upload stream (behind the kernels that last read this slot,
      // which ev_tail / ev_side cover) and the kernels wait for it when they are finally launched
      HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_tail_cur ? c->ev_tail_cur : c->ev_tail, 0));
      if (c->nb_pending) HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_nb, 0));
      HIPCHK(c, launch_phase_expand(g_inc, g_st, g_point, d_inc, d_start, d_point, batch, nchunks, LRH_PH_CHUNK, c->stream3));
      HIPCHK(c, hipEventRecord(c->ph_ev[slot], c->stream3));
      LRH_DEVICE_WORK(c, {
        HIPCHK(c, hipStreamWaitEvent(c->cur, c->ph_ev[slot], 0));
        ProfScope ps(c, "mix1");
        HIPCHK(c, launch_mix1_back(mix1_n, a, batch, c->cur));
        HIPCHK(c, launch_mix1_out(o, batch, c->cur));
      });
    } else {
      c->ph_pending[slot] = true;
      LRH_DEVICE_WORK(c, {
        HIPCHK(c, launch_phase_expand(g_inc, g_st, g_point, d_inc, d_start, d_point, batch, nchunks, LRH_PH_CHUNK, c->cur));
        HIPCHK(c, hipEventRecord(c->ph_ev[slot], c->cur));
        c->ph_pending[slot] = false;
        ProfScope ps(c, "mix1");
        HIPCHK(c, launch_mix1_back(mix1_n, a, batch, c->cur));
        HIPCHK(c, launch_mix1_out(o, batch, c->cur));
      });
    }
  } else {
    LRH_DEVICE_WORK(c, { ProfScope ps(c, "mix1"); HIPCHK(c, launch_mix1_out(o, batch, c->cur)); });
  }
  p->timf3_pa = (p->timf3_pa + batch * 2 * block2) & c->timf3_mask;      // mix1.c:991 / 1039
  return LRH_OK;
}

int lrh_fft2_mix1_fixed(lrh_ctx *c, lrh_ptrs *p, int batch)
try {
  LRH_ENTER(c);
  if (!c || !p || batch < 1 || batch > c->cfg.max_fft2n) return LRH_EINVAL;
  if (!c->cfg.second_fft_enable) return fail(c, LRH_ESTATE, "fft2_mix1_fixed needs second_fft_enable");
  int ratio = c->N2 / c->N1; if (ratio < 1) ratio = 1;
  int lim_hi = ratio * (c->N1 - 1); if (lim_hi > c->N2) lim_hi = c->N2;       // mix1.c:957 (nn*fft1_last_point)/2 bins
  int rc = mix1_run(c, p, batch, c->d_fft2, c->N2, p->fft2_nx, c->fft2n_mask, lim_hi);
  if (rc) return rc;
  p->fft2_nx = (p->fft2_nx + batch) & c->fft2n_mask;                          // mix1.c:992
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_fft2_mix1_afc(lrh_ctx *c, lrh_ptrs *p, int batch, lrh_afc *afc)
try {
  LRH_ENTER(c);
  if (!c || !p || !afc || !afc->mix1_fq_mid || !afc->mix1_fq_slope || !afc->mix1_fq_curv || !afc->mix1_fq_start || batch < 1 || batch > c->cfg.max_fft2n) return LRH_EINVAL;
  if (!c->cfg.second_fft_enable) return fail(c, LRH_ESTATE, "fft2_mix1_afc needs second_fft_enable");
  if (c->ms.mix1_selfreq < 0) return fail(c, LRH_ESTATE, "fft2_mix1_afc needs a selected frequency");
  int ratio = c->N2 / c->N1; if (ratio < 1) ratio = 1;
  int lim_hi = ratio * (c->N1 - 1); if (lim_hi > c->N2) lim_hi = c->N2;
  int rc = mix1_run(c, p, batch, c->d_fft2, c->N2, p->fft2_nx, c->fft2n_mask, lim_hi, afc, p->fft2_na);
  if (rc) return rc;
  p->fft2_nx = (p->fft2_nx + batch) & c->fft2n_mask;                          // mix1.c:931
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_fft1_mix1_afc(lrh_ctx *c, lrh_ptrs *p, int batch, lrh_afc *afc)
try {
  LRH_ENTER(c);
  if (!c || !p || !afc || !afc->mix1_fq_mid || !afc->mix1_fq_slope || !afc->mix1_fq_curv || !afc->mix1_fq_start || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  if (c->cfg.second_fft_enable) return fail(c, LRH_ESTATE, "fft1_mix1_afc needs second_fft_enable == 0");
  if (c->ms.mix1_selfreq < 0) return fail(c, LRH_ESTATE, "fft1_mix1_afc needs a selected frequency");
  int rc = mix1_run(c, p, batch, c->d_fft1, c->N1, (p->fft1_px / (2 * c->N1)) & c->fft1n_mask, c->fft1n_mask, c->N1 - 1, afc, p->fft1_nb);
  if (rc) return rc;
  p->fft1_nx = (p->fft1_nx + batch) & c->fft1n_mask;                          // mix1.c:1095-1096
  p->fft1_px = (p->fft1_px + batch * 2 * c->N1) & c->fft1_mask;
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_fft1_mix1_fixed(lrh_ctx *c, lrh_ptrs *p, int batch)
try {
  LRH_ENTER(c);
  if (!c || !p || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  if (c->cfg.second_fft_enable) return fail(c, LRH_ESTATE, "fft1_mix1_fixed needs second_fft_enable == 0");
  int rc = mix1_run(c, p, batch, c->d_fft1, c->N1, (p->fft1_px / (2 * c->N1)) & c->fft1n_mask, c->fft1n_mask, c->N1 - 1);   // mix1.c:1017-1019
  if (rc) return rc;
  p->fft1_nx = (p->fft1_nx + batch) & c->fft1n_mask;                          // mix1.c:1040-1041
  p->fft1_px = (p->fft1_px + batch * 2 * c->N1) & c->fft1_mask;
  return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- fft3 / mix2
int lrh_set_bg_filterfunc(lrh_ctx *c, const float *f)
try {
  if (!c || !f) return LRH_EINVAL;
  if (!c->N3) return fail(c, LRH_ESTATE, "fft3 not configured");
  LRH_ENTER(c);
  HIPCHK(c, stage_h2d(c, c->d_bgfilt, f, 4 * c->N3, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_make_fft3_all(lrh_ctx *c, lrh_ptrs *p, int batch)
try {
  if (!c || !p || batch < 1) return LRH_EINVAL;
  if (!c->N3) return fail(c, LRH_ESTATE, "fft3 not configured");
  if (batch > c->cfg.max_fft3n) return LRH_EINVAL;
  LRH_ENTER(c);
  LRH_WRITES(c, RB(LRH_RING_FFT3));
  Fft3Args a;
  a.timf3 = c->d_timf3; a.mask = c->cfg.timf3_size / 2 - 1; a.px_first = p->timf3_px / 2; a.step = c->M3;
  a.window = c->d_window3; a.tw = c->d_tw3; a.out = c->d_fft3;
  a.first_slot = p->fft3_pa / (2 * c->N3); a.slot_mask = c->cfg.max_fft3n - 1;
  const int fft3_n = c->cfg.fft3_n;
  LRH_DEVICE_WORK(c, { ProfScope ps(c, "fft3"); HIPCHK(c, launch_fft3(fft3_n, a, batch, c->cur)); });
  p->timf3_px = (p->timf3_px + batch * 2 * c->M3) & c->timf3_mask;                      // fft3.c:784
  p->fft3_pa = (p->fft3_pa + batch * 2 * c->N3) & (c->cfg.max_fft3n * 2 * c->N3 - 1);   // fft3.c:797
  return LRH_OK;
}
LRH_CATCH(c)

// two coupled channels (include/linrad_hip.h): polarisation transform of fft3_mix2, mix2.c:340-343, 377-380
int lrh_set_pol(lrh_ctx *c, float c1, float c2, float c3)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (c->cfg.blanker_channels != 2) return fail(c, LRH_ESTATE, "blanker_channels != 2");
  if ((c->cfg.timf1_channel_index & 1) == 0) { c->pol_wa = make_float2(c1, 0.f); c->pol_wb = make_float2(-c2, -c3); }   // A += c1 X,          B -= (c2 + j c3) X
  else                                       { c->pol_wa = make_float2(c2, -c3); c->pol_wb = make_float2(c1, 0.f); }    // A += (c2 - j c3) Y, B += c1 Y
  c->pol_set = true;
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_set_combine_weights(lrh_ctx *c, float wa_re, float wa_im, float wb_re, float wb_im)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (!c->d_xpol) return fail(c, LRH_ESTATE, "fft3 not configured");
  c->pol_wa = make_float2(wa_re, wa_im); c->pol_wb = make_float2(wb_re, wb_im); c->pol_set = true;
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_mix2_pol_begin(lrh_ctx *c, const lrh_ptrs *p, int batch, size_t *count)
try {
  if (!c || !p || !count || batch < 1) return LRH_EINVAL;
  if (!c->N3 || !c->d_xpol) return fail(c, LRH_ESTATE, "fft3 not configured");
  if (!c->pol_set) return fail(c, LRH_ESTATE, "lrh_set_pol / lrh_set_combine_weights first");
  if (batch > c->cfg.max_fft3n) return LRH_EINVAL;
  LRH_ENTER(c);
  PolArgs a;
  a.fft3 = c->d_fft3; a.n3 = c->N3; a.first_slot = p->fft3_px / (2 * c->N3); a.slot_mask = c->cfg.max_fft3n - 1; a.nm = c->Nm2; a.batch = batch;
  a.wa = c->pol_wa; a.wb = c->pol_wb;
  a.out = c->d_xpol;
  { ProfScope ps(c, "pol"); HIPCHK(c, launch_pol(a, c->cur)); }
  c->pol_batch = batch;
  *count = (size_t)4 * batch * c->Nm2;
  return LRH_OK;
}
LRH_CATCH(c)

// bg.mixer_mode = 2 (mix2.c:217-246): the FIR make_bg_filter derives from the filter function (baseb_graph.c:1560-1634), handed over like bg_filterfunc
int lrh_set_basebraw_fir(lrh_ctx *c, const float *fir, int pts)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (!c->N3 || c->pol_set) return fail(c, LRH_ESTATE, "fft3 not configured, or a coherent combine is set (the FIR decimator is the one-channel form)");
  if (fir && (pts < 1 || !(pts & 1) || pts + pts / 2 > c->I3 + c->N3 / c->Nm2 + 1)) return LRH_EINVAL;   // the first FIR of a transform must not reach behind its samples
  HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream2)); if (c->stream_nb) HIPCHK(c, hipStreamSynchronize(c->stream_nb));
  if (c->d_bbfir) { lrh_dev_free(c->d_bbfir); c->d_bbfir = nullptr; }
  c->bbfir_pts = 0;
  if (!fir) return LRH_OK;
  { const int rc_ = dev_alloc(c, &c->d_bbfir, pts, false); if (rc_) return rc_; }
  HIPCHK(c, stage_h2d(c, c->d_bbfir, fir, sizeof(float) * pts, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->bbfir_pts = pts;
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_fft3_mix2(lrh_ctx *c, lrh_ptrs *p, int batch)
try {
  if (!c || !p || batch < 1) return LRH_EINVAL;
  if (!c->N3) return fail(c, LRH_ESTATE, "fft3 not configured");
  if (batch > c->cfg.max_fft3n) return LRH_EINVAL;
  LRH_ENTER(c);
  LRH_WRITES(c, RB(LRH_RING_BASEB_RAW));
  if (c->d_bbfir) {                                        // bg.mixer_mode == 2
    Mix2FirArgs f;
    f.timf3 = c->d_timf3; f.mask = c->cfg.timf3_size / 2 - 1; f.py_first = p->timf3_py / 2; f.step = c->M3;
    f.n3 = c->N3; f.m3 = c->M3; f.nm2new = c->Mm2; f.resamp = c->N3 / c->Nm2; f.fir = c->d_bbfir; f.pts = c->bbfir_pts;
    f.baseb = c->d_baseb; f.bmask = c->cfg.baseband_size - 1; f.pa_first = p->baseb_pa;
    LRH_DEVICE_WORK(c, { ProfScope ps(c, "mix2"); HIPCHK(c, launch_mix2_fir(f, batch, c->cur)); });
    p->baseb_pa = (p->baseb_pa + batch * c->Mm2) & (c->cfg.baseband_size - 1);
    p->fft3_px = (p->fft3_px + batch * 2 * c->N3) & (c->cfg.max_fft3n * 2 * c->N3 - 1);
    p->timf3_py = (p->timf3_py + batch * 2 * c->M3) & c->timf3_mask;                     // mix2.c:2060
    return LRH_OK;
  }
  const bool pol = c->pol_set;
  if (pol && c->pol_batch != batch) return fail(c, LRH_ESTATE, "lrh_mix2_pol_begin and the all-reduce come first");
  c->pol_batch = 0;
  Mix2Args a;
  const int slot = (c->cfg.blanker_channels == 2 && (c->cfg.timf1_channel_index & 1)) ? 1 : 0;   // B for the second channel of a coupled pair
  a.pol = pol ? c->d_xpol + (size_t)slot * batch * c->Nm2 : nullptr;
  a.fft3 = c->d_fft3; a.n3 = c->N3; a.first_slot = p->fft3_px / (2 * c->N3); a.slot_mask = c->cfg.max_fft3n - 1;
  a.filt = c->d_bgfilt; a.tw = c->d_twm2; a.scratch = c->d_mix2_scratch; a.nm = c->Nm2;
  Mix1OutArgs o; memset(&o, 0, sizeof o);
  o.scratch = c->d_mix2_scratch; o.timf3 = c->d_baseb; o.mask2 = c->cfg.baseband_size - 1; o.pa_first = p->baseb_pa; o.block = c->Mm2;
  o.nm = c->Nm2; o.overlap = c->Im2 != 0; o.selected = 1; o.rotate = 0;
  if (c->Im2 != 0 && c->Im2 != c->Mm2) {                   // THIRD_FFT_SINPOW neither 0 nor 2: crossover functions (mix2.c:177-216)
    if (c->Xm2 < 1) return fail(c, LRH_EINVAL, "mix2 window without a crossover region");
    o.xover = c->Xm2; o.im = c->Im2; o.win = c->d_mix2win; o.sin2win = c->d_sin2win2; o.cos2win = c->d_cos2win2;
  }
  const int mix2_n = c->cfg.mix2_n;
  LRH_DEVICE_WORK(c, {
    ProfScope ps(c, "mix2");
    HIPCHK(c, launch_mix2_back(mix2_n, a, batch, c->cur));
    HIPCHK(c, launch_mix1_out(o, batch, c->cur));
  });
  p->baseb_pa = (p->baseb_pa + batch * c->Mm2) & (c->cfg.baseband_size - 1);             // mix2.c:1079, 2057
  p->fft3_px = (p->fft3_px + batch * 2 * c->N3) & (c->cfg.max_fft3n * 2 * c->N3 - 1);   // mix2.c:2058
  p->timf3_py = (p->timf3_py + batch * 2 * c->M3) & c->timf3_mask;                      // mix2.c:2060
  return LRH_OK;
}
LRH_CATCH(c)

// compute_timf2_powersum, wcw.c:80-138
int lrh_compute_timf2_powersum(lrh_ctx *c, lrh_ptrs *p)
try {
  LRH_ENTER(c);
  if (!c || !p) return LRH_EINVAL;
  LRH_WRITES(c, RB(LRH_RING_TIMF2_BLOCKPOWER));
  const int blk = c->cfg.timf2_blockpower_block;
  if (blk <= 0 || (blk & 3)) return fail(c, LRH_ESTATE, "timf2_blockpower_block not configured");
  const int avail = (p->timf2_pn2 - p->timf2_pb + 4 * c->cfg.timf2pow_size) & c->timf2_mask;
  int n = 0;
  if (avail > blk) n = (avail - 1) / blk;                                     // loop count of wcw.c:84
  if (n <= 0) return LRH_OK;
  BlockpowerArgs a;
  a.timf2w = c->d_timf2w; a.mask = c->timf2pow_mask; a.first = p->timf2_pb / 4; a.block = blk / 4;
  a.out = c->d_blockpower; a.out_mask = c->cfg.timf2_blockpower_size - 1; a.out_first = p->timf2_blockpower_pa;
  ProfScope ps(c, "blockpower");
  HIPCHK(c, launch_blockpower(a, n, c->cur));
  p->timf2_pb = (p->timf2_pb + n * blk) & c->timf2_mask;
  p->timf2_blockpower_pa = (p->timf2_blockpower_pa + n) & (c->cfg.timf2_blockpower_size - 1);
  return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- orchestration
// The launches the one-round-late schedule still holds when lrh_wideband_dsp returns (blanker, fft2 + mix1 + narrowband tail of its
// last round): issued by the next lrh_wideband_dsp as if the rounds had been one call, or here, by the first other entry point that
// needs the chain's results.
static int flush_pending(lrh_ctx *c)
{
  if (!c->pend) return LRH_OK;
  c->pend = false;
  hipStream_t S1 = c->stream, S2 = c->stream2;
  struct Restore { lrh_ctx *c; ~Restore() { c->cur = c->stream; c->rec = nullptr; c->split_fft2_tail = false; c->in_dsp--; c->pend_b.clear(); c->pend_t.clear(); } } restore{c};
  c->in_dsp++;
  if (c->pend_tail_flushed) HIPCHK(c, hipStreamWaitEvent(S2, c->ev_tail_cur ? c->ev_tail_cur : c->ev_tail, 0));
  c->pend_tail_flushed = false;
  c->rec = nullptr; c->cur = S2;
  for (auto &op : c->pend_b) { const int r = op(c); if (r) return r; }
  HIPCHK(c, hipEventRecord(c->ev_blank, S2));
  HIPCHK(c, hipStreamWaitEvent(S1, c->ev_blank, 0));
  c->cur = S1; c->split_fft2_tail = true;
  for (auto &op : c->pend_t) { const int r = op(c); if (r) { c->last_main_ev = nullptr; return r; } }
  c->split_fft2_tail = false; c->last_main_ev = nullptr;
  HIPCHK(c, hipEventRecord(c->ev_side, S2)); HIPCHK(c, hipStreamWaitEvent(S1, c->ev_side, 0));
  if (c->nb_pending) HIPCHK(c, hipStreamWaitEvent(S1, c->ev_nb, 0));
  return LRH_OK;
}

static void join_side_tail(lrh_ctx *c)      // later work on the main stream sees what the side-stream tails of small rounds / the narrowband stream wrote
{
  if (c->st_pending) (void)hipStreamWaitEvent(c->stream, c->ev_st[(c->st_n - 1) & 1], 0);
  c->st_pending = false;
  if (c->nb_join_pending && c->nb_pending) (void)hipStreamWaitEvent(c->stream, c->ev_nb, 0);
  c->nb_join_pending = false;
}

// single-CPU branch of wideband_dsp (wcw.c:1036-1118), `batch` fft1 blocks per round.
// Device schedule when several rounds are requested (second fft on): the transform kernels of N = 16384 occupy one
// workgroup per CU and are latency/compute bound, while fft1_c's sums, the blanker and the fft2 power sums are short
// bandwidth-bound kernels that need no LDS -- so the latter run on a second stream underneath the former:
//   main:  fft1(k+1) | fft2(k) mix1(k) | timf2(k+1) | fft1(k+2) ...
//   side:  blanker(k) sumsq(k+1) slowsum(k+1) | powersum2(k) waterfall(k) | ...
// Events carry exactly the data dependencies of the serial order; host bookkeeping is unchanged.
// The narrowband side behind mix1 (wcw.c:1788, 1828: EVENT_FFT3 -> do_fft3, fft3.c:35-60 -> EVENT_MIX2 -> do_mix2, mix2.c:41-80):
// every fft3 transform timf3 holds, then its filter / decimate step.  With a coherent combine configured (lrh_set_pol /
// lrh_set_combine_weights) the caller's collective sits between lrh_mix2_pol_begin and lrh_fft3_mix2, so the caller runs
// this part itself.
static int narrow_tail(lrh_ctx *c, lrh_ptrs *p)
{
  if (!c->N3 || c->pol_set || c->ms.mix1_selfreq < 0) return LRH_OK;
  int rc;
  const int have = (p->timf3_pa - p->timf3_px + c->cfg.timf3_size) & c->timf3_mask;       // fft3.c:54-55
  int k = have < 2 * c->N3 ? 0 : 1 + (have - 2 * c->N3) / (2 * c->M3);
  const int cap = c->cfg.max_fft3n / 2 > 0 ? c->cfg.max_fft3n / 2 : 1;                    // producer and consumer share the fft3 ring
  while (k > 0) {
    const int kb = k < cap ? k : cap;
    if ((rc = lrh_make_fft3_all(c, p, kb))) return rc;
    if ((rc = lrh_fft3_mix2(c, p, kb))) return rc;
    k -= kb;
  }
  return LRH_OK;
}

static int round_tail(lrh_ctx *c, lrh_ptrs *p)      // fft2 + mix1 (+ fft3 / mix2) for everything the blanker has released
{
  int rc;
  const int avail = (p->timf2_pn2 - p->timf2_px + 4 * c->cfg.timf2pow_size) & c->timf2_mask;   // wcw.c:265-266
  int k = 0;
  if (avail >= 4 * c->N2) k = 1 + (avail - 4 * c->N2) / (4 * c->M2);
  while (k > 0) {
    const int kb = k < c->cfg.max_fft2n ? k : c->cfg.max_fft2n;
    // (the stream switches are part of the recorded work: a parked round is issued -- and the split schedule known -- a round later)
    // the narrowband kernels of the previous group may still read the fft2 slots this group overwrites (they have had a round's time)
    int wait_idx = -1;
    {
      const unsigned kept = c->nb_seq < 4 ? c->nb_seq : 4;
      for (unsigned dd = 1; dd <= kept; dd++) {
        const int j = (int)((c->nb_seq - dd) & 3);
        if (c->f2_total + kb - c->nb_first_total[j] > (long)c->cfg.max_fft2n) { wait_idx = j; break; }     // the newest group whose slots these transforms reach
      }
      if (wait_idx < 0 && c->nb_seq > 4) wait_idx = (int)(c->nb_seq & 3);      // groups no longer tracked: the oldest kept event is behind them
      static const bool newest = getenv("LRH_NB_WAIT_NEWEST") && atoi(getenv("LRH_NB_WAIT_NEWEST"));     // (comparison: always the newest group, as before)
      if (newest && c->nb_seq > 0) wait_idx = (int)((c->nb_seq - 1) & 3);
    }
    const int rec_idx = (int)(c->nb_seq & 3);
    c->nb_first_total[rec_idx] = c->f2_total; c->nb_seq++; c->f2_total += kb;
    LRH_DEVICE_WORK(c, { if (wait_idx >= 0 && c->nb_ev_valid[wait_idx] && c->split_fft2_tail && c->nb_split) HIPCHK(c, hipStreamWaitEvent(c->cur, c->ev_nb_ring[wait_idx], 0)); });
    if ((rc = lrh_make_fft2(c, p, kb))) return rc;
    LRH_DEVICE_WORK(c, {
      if (c->split_fft2_tail && c->nb_split) {
        if (c->last_main_ev) HIPCHK(c, hipStreamWaitEvent(c->stream_nb, c->last_main_ev, 0));      // make_fft2 has just recorded ev_fft2 there
        else { HIPCHK(c, hipEventRecord(c->ev_f2done, c->cur)); HIPCHK(c, hipStreamWaitEvent(c->stream_nb, c->ev_f2done, 0)); }
        c->nb_keep = c->cur; c->cur = c->stream_nb;
      } else c->last_main_ev = nullptr;                     // the narrowband kernels follow on the main stream
      });
    rc = lrh_fft2_mix1_fixed(c, p, kb);
    if (!rc) rc = narrow_tail(c, p);
    if (rc && !c->rec && c->nb_keep) { c->cur = c->nb_keep; c->nb_keep = nullptr; }
    if (rc) return rc;
    LRH_DEVICE_WORK(c, {
      if (c->nb_keep) {
        HIPCHK(c, hipEventRecord(c->ev_nb_ring[rec_idx], c->stream_nb)); c->nb_ev_valid[rec_idx] = true; c->ev_nb = c->ev_nb_ring[rec_idx]; c->nb_pending = true;
        c->cur = c->nb_keep; c->nb_keep = nullptr;
      } });
    if (rc) return rc;
    k -= kb;
  }
  return LRH_OK;
}

static void advance_fft1(lrh_ctx *c, lrh_ptrs *p, int B)     // caller-side pointers, wcw.c:1037-1047
{
  const int C = c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1;
  p->timf1p_px = (p->timf1p_px + B * c->M1 * (c->cfg.timf1_dword_input ? 8 : 4) * C) & c->timf1_bytemask;
  p->fft1_pa = (p->fft1_pa + B * 2 * c->N1) & c->fft1_mask;
  p->fft1_na = p->fft1_pa / (2 * c->N1);
  p->fft1_nm = p->fft1_nm + B > c->fft1n_mask ? c->fft1n_mask : p->fft1_nm + B;
}

// ---- correlation spectrum of two coupled channels (include/linrad_hip.h): fft1_corrsum, fft1_slowcorr, fft1_slowcorr_tot
int lrh_set_correlation(lrh_ctx *c, int on)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (c->cfg.blanker_channels != 2) return fail(c, LRH_ESTATE, "blanker_channels != 2");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (void **q_ : { (void **)&c->d_xspec, (void **)&c->d_corrsum, (void **)&c->d_slowcorr, (void **)&c->d_slowcorr_tot }) if (*q_) { lrh_dev_free(*q_); *q_ = nullptr; }
  c->corr_on = false; c->slowcorr_tot_avgnum = 0;
  if (!on) return LRH_OK;
  int rc;
  if ((rc = dev_alloc(c, &c->d_xspec, (size_t)2 * c->cfg.max_batch * c->N1, false)) || (rc = dev_alloc(c, &c->d_corrsum, (size_t)c->cfg.fft1_sumsq_bufsize)) ||
      (rc = dev_alloc(c, &c->d_slowcorr, (size_t)c->N1)) || (rc = dev_alloc(c, &c->d_slowcorr_tot, (size_t)c->N1))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->corr_on = true;
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_get_slowcorr_tot_avgnum(lrh_ctx *c, int *n) try { LRH_LOCK(c); if (!c || !n) return LRH_EINVAL; *n = c->slowcorr_tot_avgnum; return LRH_OK; } LRH_CATCH(c)
int lrh_fft1_corr_begin(lrh_ctx *c, const lrh_ptrs *at, int batch, size_t *count)
try {
  LRH_ENTER(c);
  if (!c || !at || !count || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  if (!c->corr_on) return fail(c, LRH_ESTATE, "lrh_set_correlation first");
  { const int rc_ = join_handles(c); if (rc_) return rc_; }
  const int N = c->N1, nb = at->fft1_nb & c->fft1n_mask;
  float2 *slot = c->d_xspec + (size_t)(c->cfg.timf1_channel_index & 1) * batch * N;
  const int first = std::min(batch, c->cfg.max_fft1n - nb);            // the ring span may wrap once
  HIPCHK(c, hipMemcpyAsync(slot, c->d_fft1 + (size_t)nb * N, (size_t)first * N * sizeof(float2), hipMemcpyDeviceToDevice, c->cur));
  if (batch > first) HIPCHK(c, hipMemcpyAsync(slot + (size_t)first * N, c->d_fft1, (size_t)(batch - first) * N * sizeof(float2), hipMemcpyDeviceToDevice, c->cur));
  *count = (size_t)batch * 2 * N;
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_fft1_corr_finish(lrh_ctx *c, const lrh_ptrs *at, int batch)
try {
  LRH_ENTER(c);
  LRH_WRITES(c, RB(LRH_RING_FFT1_CORRSUM) | RB(LRH_RING_FFT1_SLOWCORR) | RB(LRH_RING_FFT1_SLOWCORR_TOT));
  if (!c || !at || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  if (!c->corr_on) return fail(c, LRH_ESTATE, "lrh_set_correlation first");
  const int N = c->N1;
  CorrArgs a; memset(&a, 0, sizeof a);
  a.x = c->d_xspec; a.y = c->d_xspec + (size_t)batch * N; a.n = N; a.batch = batch;
  a.corrsum = c->d_corrsum; a.sumsq_mask = c->sumsq_mask; a.avg = c->cfg.fft_avg1num; a.c0 = at->fft1_sumsq_counter; a.pa0 = at->fft1_sumsq_pa;
  a.slowcorr = c->d_slowcorr; a.tot = c->d_slowcorr_tot; a.bufsize = c->cfg.fft1_sumsq_bufsize; a.avg2 = c->cfg.fft_avg2num;
  a.nupd = (at->fft1_sumsq_counter + batch) / a.avg; a.recalc0 = at->fft1_sumsq_recalc; a.step = c->cfg.wg_xpoints / c->cfg.slowsum_fresh_recalc;
  { ProfScope ps(c, "corrsum"); HIPCHK(c, launch_corrsum(a, c->cur)); }
  c->slowcorr_tot_avgnum += a.nupd * a.avg;
  return LRH_OK;
}
LRH_CATCH(c)

int lrh_set_exchange(lrh_ctx *c, lrh_exchange_fn fn, void *user)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  if (c->cfg.blanker_channels != 2) return fail(c, LRH_ESTATE, "blanker_channels != 2");
  c->xfn = fn; c->xuser = user;
  return LRH_OK;
}
LRH_CATCH(c)
// one exchange point: everything that fills the buffer is on the main stream by now; the caller's function puts the collective there too
static int exchange(lrh_ctx *c, int which, int op, size_t count, const void *own = nullptr)
{
  if (!count) return LRH_OK;
  void *ptr = nullptr;
  { const int rc = lrh_exchange_ptr(c, which, &ptr); if (rc) return rc; }
  if (c->xfn(c->xuser, which, op, ptr, count, (void *)c->stream, own) != 0) return fail(c, LRH_EDEVICE, "the registered exchange function failed");
  return LRH_OK;
}
static int narrow_tail(lrh_ctx *c, lrh_ptrs *p);
// Two coupled channels: the single-CPU order of wideband_dsp with the cross-channel exchanges at their places (include/linrad_hip.h,
// lrh_set_exchange).  Everything is enqueued on the main stream -- the exchanges order the two ranks' streams against each other, so the
// side-stream overlap of the single-channel schedules has no place here -- and nothing waits on the host (the linear blanker's resume
// point excepted, see lrh_first_noise_blanker).
static int dsp_coupled(lrh_ctx *c, lrh_ptrs *p, int nblocks, int batch)
{
  int rc;
  const bool fuse = c->fuse_sumsq && c->timf2_mode == 1 && c->d_ss_part;
  struct FuseGuard { lrh_ctx *c; ~FuseGuard() { c->ss_defer = false; c->ss_queue.clear(); c->f1_defer = false; if (c->f1_have) launch_parked_fft1(c); c->in_dsp--; } } guard{c};
  c->in_dsp++;
  c->f1_defer = fuse && c->fuse_fft1 && !c->fft1_big && (c->cfg.fft1_n == 14 || (c->fuse_v && c->cfg.fft1_n >= 12 && c->cfg.fft1_n <= 13));
  while (nblocks > 0) {
    const int B = nblocks < batch ? nblocks : batch;
    if ((rc = lrh_fft1_b(c, 0, p->timf1p_px, p->fft1_pa, B))) return rc;
    advance_fft1(c, p, B);
    const lrh_ptrs at1 = *p;
    c->ss_defer = fuse; rc = lrh_fft1_c(c, p, B); c->ss_defer = false;
    if (rc) return rc;
    if ((rc = lrh_make_timf2(c, p, B))) return rc;
    { std::vector<std::function<int(lrh_ctx *)>> q; q.swap(c->ss_queue); for (auto &op : q) if ((rc = op(c))) return rc; }
    if (c->corr_on) {                                   // the spectra are in the ring by now (k_fft1w stores every bin in this mode)
      size_t ns = 0;
      if ((rc = lrh_fft1_corr_begin(c, &at1, B, &ns)) || (rc = exchange(c, LRH_X_SPEC, LRH_XOP_GATHER, ns)) || (rc = lrh_fft1_corr_finish(c, &at1, B))) return rc;
    }
    int cnt = 0;
    if ((rc = lrh_blanker_begin(c, p, &cnt))) return rc;
    if (cnt > 0) {
      if ((rc = exchange(c, LRH_X_PWR, LRH_XOP_SUM, (size_t)cnt))) return rc;
      size_t nw = 0;
      if ((rc = lrh_blanker_weak_span(c, &nw))) return rc;
      if (nw && (rc = exchange(c, LRH_X_WEAK, LRH_XOP_GATHER, nw))) return rc;
    }
    if ((rc = lrh_first_noise_blanker(c, p))) return rc;
    if (cnt > 0) {
      if ((rc = exchange(c, LRH_X_STAT, LRH_XOP_SUM, 2))) return rc;
      if ((rc = lrh_blanker_finish(c, p))) return rc;
    }
    const int avail = (p->timf2_pn2 - p->timf2_px + 4 * c->cfg.timf2pow_size) & c->timf2_mask;   // wcw.c:265-266
    int k = avail >= 4 * c->N2 ? 1 + (avail - 4 * c->N2) / (4 * c->M2) : 0;
    while (k > 0) {
      const int kb = k < c->cfg.max_fft2n ? k : c->cfg.max_fft2n;
      const lrh_ptrs at = *p;
      size_t n = 0;
      if ((rc = lrh_make_fft2(c, p, kb))) return rc;
      // the batch's transforms as the collective's send buffer where they lie in the fft2 ring (no copy into the own slot: 537 MB per
      // round of 4096 blocks), unless the span wraps around the ring
      const int na_ = at.fft2_na & c->fft2n_mask;
      if (na_ + kb <= c->cfg.max_fft2n) {
        c->xy_own_src = c->d_fft2 + (size_t)na_ * c->N2; n = (size_t)kb * 2 * c->N2;
        rc = exchange(c, LRH_X_BINS, LRH_XOP_GATHER, n, c->xy_own_src);
      } else if (!(rc = lrh_fft2_xy_begin(c, &at, kb, &n))) rc = exchange(c, LRH_X_BINS, LRH_XOP_GATHER, n);
      if (!rc) rc = lrh_fft2_xy_finish(c, &at, kb);
      c->xy_own_src = nullptr;
      if (rc || (rc = lrh_fft2_mix1_fixed(c, p, kb))) return rc;
      if (c->N3 && c->ms.mix1_selfreq >= 0) {
        if (!c->pol_set) { if ((rc = narrow_tail(c, p))) return rc; }
        else {
          const int have = (p->timf3_pa - p->timf3_px + c->cfg.timf3_size) & c->timf3_mask;
          int k3 = have < 2 * c->N3 ? 0 : 1 + (have - 2 * c->N3) / (2 * c->M3);
          const int cap = c->cfg.max_fft3n / 2 > 0 ? c->cfg.max_fft3n / 2 : 1;
          while (k3 > 0) {
            const int k3b = k3 < cap ? k3 : cap;
            size_t np = 0;
            if ((rc = lrh_make_fft3_all(c, p, k3b)) || (rc = lrh_mix2_pol_begin(c, p, k3b, &np)) || (rc = exchange(c, LRH_X_POL, LRH_XOP_SUM, np)) ||
                (rc = lrh_fft3_mix2(c, p, k3b))) return rc;
            k3 -= k3b;
          }
        }
      }
      k -= kb;
    }
    nblocks -= B;
  }
  return LRH_OK;
}

int lrh_wideband_dsp(lrh_ctx *c, lrh_ptrs *p, int nblocks, int batch)
try {
  if (!c || !p || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  LRH_LOCK(c);
  // read-backs still out: this call's kernels go to several streams -- the host waits for the copies (a few hundred KB each, started by an earlier call)
  if (c->stream_out && !c->in_dsp) for (int i_ = 0; i_ < LRH_NOUT; i_++) if (c->out_busy[i_]) HIPCHK(c, hipEventSynchronize(c->ev_out_done[i_]));
  if (c->cfg.blanker_channels == 2) {
    if (!c->xfn) return fail(c, LRH_ESTATE, "two coupled channels: register the exchange function (lrh_set_exchange) or make the stage calls with the exchanges between them (lrh_blanker_begin)");
    if (c->pend) { const int rcf_ = flush_pending(c); if (rcf_) return rcf_; }
    return dsp_coupled(c, p, nblocks, batch);
  }
  struct HostTimer { lrh_ctx *c; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); double cpu0 = thread_cpu_ms();
                     static double thread_cpu_ms() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
                     ~HostTimer() { c->host_ms_dsp += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); c->host_n_dsp++;
                                    c->host_cpu_ms_dsp += thread_cpu_ms() - cpu0; } } host_timer{c};
  int rc;
  // a call that fails between parking the linear blanker's search and picking its result up must not leave the search marked as parked:
  // every later blanker call would answer LRH_ESTATE
  // later API calls are ordered on the main stream: the entry point that comes next orders it behind the narrowband stream (LRH_ENTER) -- not this
  // call's end, or a caller that passes one round per call would have every call's first transform wait for the previous call's last mix1 / fft3 / mix2
  struct NbJoin { lrh_ctx *c; ~NbJoin() { if (c->nb_pending) c->nb_join_pending = true; } } nb_join{c};
  struct ClvGuard { lrh_ctx *c; bool ok = false; ~ClvGuard() { if (!ok) { c->clv_wait = false; c->clv_issued = false; } } } clv_guard{c};
  struct InDsp { lrh_ctx *c; InDsp(lrh_ctx *c_) : c(c_) { c->in_dsp++; } ~InDsp() { c->in_dsp--; } };
  // Small rounds are bound by the host's launches (~100 us per round), not by the kernels: the plain serial order has the
  // fewest stream operations and wins there (Msamples/s serial / lagged, fft1_size 16384: 82 / 80 at 1 block per round,
  // 2170 / 1990 at 32, 9420 / 9100 at 256); the two-stream schedules pay off from ~3 M samples per round (15200 / 17000 at 512).
  const bool small_rounds = !c->pipeline_forced && (long)batch * c->M1 < (3L << 20);
  const bool piped = c->pipeline && !small_rounds && c->cfg.second_fft_enable && nblocks > batch && (!c->prof || c->prof_keep_schedule) &&
                     !c->clever_on;                // the linear blanker reads its resume point back: serial schedule
  // fft1_c's sums ride inside make_timf2's kernel: fft1_c parks, make_timf2 picks up, the slow average follows
  const bool fuse15 = c->fft1_big && c->cfg.fft1_n == 15 && c->fuse_fft1;     // fft1_size 32768: k_fft1r_t2c needs no scratch of split groups
  const bool fuse = c->fuse_sumsq && c->cfg.second_fft_enable && c->timf2_mode == 1 && (c->d_ss_part || fuse15);
  // ... and fft1_b's transform rides there too (k_fft1w): parked by lrh_fft1_b, taken by lrh_make_timf2, issued as k_fft1 by whoever else reads the ring
  struct FuseGuard { lrh_ctx *c; ~FuseGuard() { c->ss_defer = false; c->ss_queue.clear(); c->f1_defer = false; if (c->f1_have) launch_parked_fft1(c); } } fuse_guard{c};
  // (a call of a few blocks is a chain of single-workgroup latencies: there one block through k_fft1w -- forward and back transform in
  // one 512-thread workgroup -- takes longer than through k_fft1 and k_timf2 one after the other: 94 against 84 us per call of 1 block,
  // 149 against 127 at 4, even at 64; the fused kernel from 32 blocks per round)
  c->f1_defer = fuse && c->fuse_fft1 && ((!c->fft1_big && (c->cfg.fft1_n == 14 || (c->fuse_v && c->cfg.fft1_n >= 12 && c->cfg.fft1_n <= 13)) && (batch >= 32 || c->fuse_fft1_forced || c->cfg.fft1_float_sparse)) || fuse15);   // (a sparse ring has no full spectrum for k_timf2's overlap: always fused)
  auto sums = [&](int B) -> int {                // fft1_c: launches at once, or parked for the next make_timf2
    c->ss_defer = fuse; const int r = lrh_fft1_c(c, p, B); c->ss_defer = false; return r;
  };
  auto sums_follow = [&](hipStream_t st) -> int {   // what fft1_c left for after make_timf2 (join, slow average)
    hipStream_t keep = c->cur; c->cur = st;
    std::vector<std::function<int(lrh_ctx *)>> q; q.swap(c->ss_queue);
    int r = LRH_OK;
    for (auto &op : q) if ((r = op(c))) break;
    c->cur = keep; return r;
  };
  // the limiter calls at the end of a pass of the reference's loop (wcw.c:1124-1133), when lrh_wideband_limiter installed them
  // (cnt1 / sumsq_pa: fft1_c's counters as they stood after THIS round's sums -- schedule 1 has already booked the next round's)
  // The first limiter only needs the round's power sums, so it is queued right behind them on the side stream -- queued at the end of
  // the round it would sit behind the waterfall, i.e. behind fft2, and the next round's make_timf2 would wait 370 us for its table.
  auto limiter1 = [&](int cnt1, int sumsq_pa, int sumsq_counter) -> int {
    if (!c->wl_on || !c->cfg.second_fft_enable || cnt1 == c->wl_cnt1) return LRH_OK;
    std::vector<std::function<int(lrh_ctx *)>> *keep_rec = c->rec; hipStream_t keep_cur = c->cur;
    c->rec = nullptr; c->cur = c->stream;
    lrh_ptrs at = *p; at.fft1_sumsq_pa = sumsq_pa; at.fft1_sumsq_counter = sumsq_counter;
    const int r = sellim_run(c, &at, &c->wl_par, 1); c->wl_cnt1 = cnt1;
    c->rec = keep_rec; c->cur = keep_cur;
    return r;
  };
  auto limiter2 = [&]() -> int {
    if (!c->wl_on || !c->wl_fft2 || !c->cfg.second_fft_enable || p->fft2_liminfo_cnt == c->wl_cnt2) return LRH_OK;
    std::vector<std::function<int(lrh_ctx *)>> *keep_rec = c->rec; hipStream_t keep_cur = c->cur;
    c->rec = nullptr; c->cur = c->stream;
    const int r = sellim_run(c, p, &c->wl_par, 2); c->wl_cnt2 = p->fft2_liminfo_cnt;
    c->rec = keep_rec; c->cur = keep_cur;
    return r;
  };
  // the one-round-late schedule, also for a single round per call when the previous call left its last round parked (or may park this one)
  const long need2 = 2L * batch * c->M1 + 2L * c->N2 + c->cfg.blnfit_range + 4L * (c->cfg.blanker_pulsewidth + 2);
  const bool lag_ok = c->pipeline >= 2 && !small_rounds && c->cfg.second_fft_enable && (!c->prof || c->prof_keep_schedule) && !(c->clever_on && c->cfg.blanker_channels == 2) &&
                      need2 <= c->cfg.timf2pow_size && (long)batch * c->M1 / c->M2 + 2 <= c->cfg.max_fft2n &&
                      !(c->wl_on && c->wl_fft2);          // the second limiter reads the power sums of this round's fft2
  // (the linear blanker's resume point comes back from the device a round late: nothing of it is carried from call to call)
  const bool carry_ok = c->persist && !c->clever_on;
  const bool lagged = lag_ok && (nblocks >= 3 * batch || (carry_ok && nblocks % batch == 0 && nblocks >= batch));
  if (c->pend && !(lagged && carry_ok && batch == c->pend_batch)) { if ((rc = flush_pending(c))) return rc; }
  InDsp in_dsp{c};
  // ... where the tail comes up every other call at most (the blanker's rate limit, blank1.c:712-715).  With a tail per call the side stream
  // is the longer of the two and the hand-over only adds to it: 140 against 128 us per call of 4 blocks, 174 against 146 at 16; 63 against 80 at 1.
  const bool side_tail = c->st_on && !piped && !lagged && small_rounds && c->cfg.second_fft_enable && !c->clever_on && !c->prof && !(c->wl_on && c->wl_fft2) &&
                         c->cfg.stupid_bln_mode != 0 && 2L * batch * c->M1 <= c->cfg.blanker_min_points && nblocks == batch;
  {   // a schedule that does not continue the previous call's side work first orders the main stream behind it (the one-round-late schedule
      // orders itself against the narrowband stream by events; the side-stream tails of small rounds follow one another on their stream)
    const bool keep_nb = lagged && c->nb_join_pending, keep_st = side_tail && c->st_pending;
    if (keep_nb) c->nb_join_pending = false;
    if (keep_st) c->st_pending = false;
    join_side_tail(c);
    c->nb_join_pending = keep_nb; c->st_pending = keep_st;
  }
  if (!piped && !lagged) {
    while (nblocks > 0) {
      const int B = nblocks < batch ? nblocks : batch;
      if ((rc = lrh_fft1_b(c, 0, p->timf1p_px, p->fft1_pa, B))) return rc;
      advance_fft1(c, p, B);
      if ((rc = c->cfg.second_fft_enable ? sums(B) : lrh_fft1_c(c, p, B))) return rc;
      if (!c->cfg.second_fft_enable) {           // wcw.c:1049-1081: fft1_c, then the narrowband side's fft1_mix1_fixed
        if ((rc = lrh_fft1_mix1_fixed(c, p, B))) return rc;
        nblocks -= B;
        continue;
      }
      if ((rc = lrh_make_timf2(c, p, B))) return rc;
      if ((rc = sums_follow(c->cur))) return rc;
      // The first limiter needs this round's sums and must not rewrite the table under make_timf2: both are on the stream by now, so
      // it goes out here on its own stream and runs beside the blanker, fft2 and mix1 (a small round is half limiter otherwise:
      // 224 -> 165 us per call of 4 blocks, 298 -> 182 at 64).  Not with the linear blanker, which reads the amplitude factor the limiter updates.
      const bool early1 = !c->clever_on;
      if (early1 && (rc = limiter1(p->fft1_liminfo_cnt, p->fft1_sumsq_pa, p->fft1_sumsq_counter))) return rc;
      if (side_tail) {
        // One block per call: fft1 + timf2 take 45 us, the blanker's five launches and fft2 + mix1 (every fourth call: blanker_min_points,
        // fft2's 32768 new samples) another 90.  Nothing the next rounds' fft1 / timf2 / sums touch: timf2 writes above timf2_pa, the
        // blanker and fft2 work below it.  The stage functions do their bookkeeping now and park the launches (as in schedule 2).
        std::vector<std::function<int(lrh_ctx *)>> qs;
        // (mix1's phase table goes up in stream order on the side stream: the upload stream of schedule 2 is ordered by events this path does not record)
        const bool eu = c->early_upload; c->early_upload = false;
        c->rec = &qs; rc = lrh_first_noise_blanker(c, p); if (!rc) rc = round_tail(c, p); c->rec = nullptr;
        c->early_upload = eu;
        if (rc) return rc;
        if (!qs.empty()) {
          hipStream_t S2 = c->stream2;
          HIPCHK(c, hipStreamWaitEvent(S2, c->ev_timf2_done, 0));                          // recorded at the end of lrh_make_timf2
          if (c->st_n >= 1) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_st[(c->st_n - 1) & 1], 0));   // the main stream stays within one tail of the side stream (rings)
          c->cur = S2;
          for (auto &op : qs) if ((rc = op(c))) break;
          c->cur = c->stream;
          if (rc) return rc;
          HIPCHK(c, hipEventRecord(c->ev_st[c->st_n & 1], S2)); c->st_n++; c->st_pending = true;
        }
        nblocks -= B;
        continue;
      }
      if ((rc = lrh_first_noise_blanker(c, p))) return rc;
      if ((rc = round_tail(c, p))) return rc;
      if ((!early1 && (rc = limiter1(p->fft1_liminfo_cnt, p->fft1_sumsq_pa, p->fft1_sumsq_counter))) || (rc = limiter2())) return rc;
      nblocks -= B;
    }
    clv_guard.ok = true; return LRH_OK;
  }
  // ---- two-stream schedules
  hipStream_t S1 = c->stream, S2 = c->stream2;
  auto on = [&](hipStream_t s) { c->cur = s; };
  struct Restore { lrh_ctx *c; ~Restore() { c->cur = c->stream; c->rec = nullptr; c->split_fft2_tail = false; } } restore{c};
  // Schedule 2 (lagged): the side kernels only find room next to k_timf2 (the two forward-transform kernels fill the
  // register file), so the blanker, fft2 and mix1 of a round are enqueued one round late:
  //   main:  timf2(k) | fft1(k+1) | fft2(k-1) mix1(k-1) | timf2(k+1) ...
  //   side:  sumsq(k) slowsum(k) blanker(k-1) | waterfall(k-1) | sumsq(k+1) ...
  // The stage functions do their pointer bookkeeping in the reference's order and park their launches in a queue
  // (LRH_DEVICE_WORK); results are those of the serial order because every ring holds two rounds (checked here).
  if (lagged) {
    struct ClvDefer { lrh_ctx *c; ClvDefer(lrh_ctx *c_) : c(c_) { c->clv_defer = c->clever_on; c->clv_amp_seq = 0; } ~ClvDefer() { c->clv_defer = false; } } clv_defer{c};
    std::vector<std::function<int(lrh_ctx *)>> qb, qt;     // parked launches: blanker / fft2+mix1 of the previous round
    const bool carry = c->pend;                            // ... which may come from the previous call
    if (carry) { qb.swap(c->pend_b); qt.swap(c->pend_t); c->pend = false; }
    auto flush = [&](std::vector<std::function<int(lrh_ctx *)>> &q, hipStream_t st) -> int {
      c->rec = nullptr; c->cur = st;
      for (auto &op : q) { const int r = op(c); if (r) { q.clear(); return r; } }
      q.clear(); return LRH_OK;
    };
    HIPCHK(c, hipEventRecord(c->ev_side, S1)); HIPCHK(c, hipStreamWaitEvent(S2, c->ev_side, 0));
    HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_side, 0));
    int left = nblocks, round = 0;
    int B = left < batch ? left : batch;
    bool have_prev = carry, tail_flushed = carry && c->pend_tail_flushed;
    on(S1); if ((rc = lrh_fft1_b(c, 0, p->timf1p_px, p->fft1_pa, B))) return rc;
    advance_fft1(c, p, B);
    if (!fuse) HIPCHK(c, hipEventRecord(c->ev_fft1, S1));
    // The threshold blanker only writes from where its span begins (it clears runs and the samples AHEAD of them, blank1.c:1049-1083),
    // which is where the previous call's release point timf2_pn2 lies, and fft2 reads below that point: blanker(k) needs nothing of
    // fft2(k-1) and goes out as soon as timf2(k) is on the stream.  It then has fft2(k-1) and fft1(k+1) + timf2(k+1) to finish in -- queued
    // a round later it ran beside fft1 + timf2 alone and took 460 of their 440 us (80 us with the chip to itself), so fft2 waited for it
    // every round.  The linear blanker's fit reaches back across its span's start and its launches wait for the host: a round late, as before.
    static const bool early_env = !(getenv("LRH_BLANK_EARLY") && !atoi(getenv("LRH_BLANK_EARLY")));
    const bool early_blank = early_env && !c->clever_on;
    hipEvent_t ev_bl_tail = c->ev_blank, ev_bl_next = c->ev_blank;
    auto side_blanker = [&]() -> int {        // blanker(k-1): after timf2(k-1) wrote and fft2(k-2) read its neighbourhood
      if (tail_flushed && !early_blank) HIPCHK(c, hipStreamWaitEvent(S2, c->ev_tail_cur ? c->ev_tail_cur : c->ev_tail, 0));
      const int r = flush(qb, S2); if (r) return r;
      HIPCHK(c, hipEventRecord(c->ev_blank, S2));
      ev_bl_tail = c->ev_blank;
      return LRH_OK;
    };
    auto main_tail = [&]() -> int {           // fft2(k-1) + mix1(k-1) on the blanked data
      HIPCHK(c, hipStreamWaitEvent(S1, ev_bl_tail, 0));
      c->split_fft2_tail = true; c->last_main_ev = nullptr;
      const int r = flush(qt, S1);
      c->split_fft2_tail = false;
      if (r) { c->last_main_ev = nullptr; return r; }
      if (c->last_main_ev) c->ev_tail_cur = c->last_main_ev;     // fft2's own event is the last thing on the main stream
      else { HIPCHK(c, hipEventRecord(c->ev_tail, S1)); c->ev_tail_cur = c->ev_tail; }
      c->last_main_ev = nullptr; tail_flushed = true;
      return LRH_OK;
    };
    while (left > 0) {
      const int Bnext = (left - B) < batch ? (left - B) : batch;
      // side: the blanker first -- fft2(k-1) on the main stream waits for it, the sums have a whole round of slack
      if (have_prev && (!early_blank || round == 0) && (rc = side_blanker())) return rc;
      // Linear blanker: its search -- a few thousand one-wave workgroups waiting on memory -- takes 0.2 ms with the chip to itself and
      // 0.55 ms beside k_fft1w, whose workgroups fill every CU's registers.  Tried and dropped: a stream confined to a share of the CUs
      // (22.0 against 26.5 Gsamples/s), and holding the transform back until the search is through (this switch: 24.0 against 26.3 --
      // the host then waits with nothing queued behind the search).
      if (have_prev && c->clv_first && c->clv_wait && c->clv_issued) HIPCHK(c, hipStreamWaitEvent(S1, c->ev_clv, 0));
      // (blanker issued at once: its wait is queued before lrh_make_timf2 records ev_timf2_done again, so that event serves)
      hipEvent_t ev_t2 = early_blank ? c->ev_timf2_done : (round & 1 ? c->ev_timf2b : c->ev_timf2);
      if (!fuse) { on(S2); HIPCHK(c, hipStreamWaitEvent(S2, c->ev_fft1, 0)); }
      if ((rc = sums(B))) return rc;
      on(S1); if ((rc = lrh_make_timf2(c, p, B))) return rc;
      if (!early_blank) HIPCHK(c, hipEventRecord(ev_t2, S1));
      // With the limiter in the call the sums' join and the slow average feed it and sit on the path to the next make_timf2: beside
      // k_fft1 (which fills every register file) they would wait for it to end, so they go first on the main stream (26 us there)
      // when the main stream's work between two make_timf2 is short enough for that wait to show (single-kernel fft2).
      // ... and always with the linear blanker: on the side stream the join and the slow average (which wait for this round's make_timf2)
      // would stand between the previous round's search and the rest of its blanker call, which the host issues only when the search
      // has reported back -- and fft2 of that round waits for exactly that (1284 -> 1249 us per round, 26.1 -> 26.9 Gsamples/s).
      hipStream_t Ss = (fuse && ((c->wl_on && c->sums_on_main) || c->clever_on)) ? S1 : S2;
      if (fuse) { if (Ss == S2) HIPCHK(c, hipStreamWaitEvent(S2, ev_t2, 0)); if ((rc = sums_follow(Ss))) return rc; }
      HIPCHK(c, hipEventRecord(c->ev_sumsq[round & 1], Ss));
      if (early_blank && Ss == S2) c->last_main_ev = c->ev_timf2_done;   // nothing on the main stream since lrh_make_timf2 recorded it
      rc = limiter1(p->fft1_liminfo_cnt, p->fft1_sumsq_pa, p->fft1_sumsq_counter);
      c->last_main_ev = nullptr;
      if (rc) return rc;
      // bookkeeping of blanker(k): its launches wait for timf2(k) and are issued in the next round
      qb.push_back([ev_t2](lrh_ctx *c) -> int { HIPCHK(c, hipStreamWaitEvent(c->stream2, ev_t2, 0)); return LRH_OK; });
      c->clv_ev_t2 = ev_t2;
      c->rec = &qb; rc = lrh_first_noise_blanker(c, p); c->rec = nullptr;
      c->clv_ev_t2 = nullptr;
      if (rc) return rc;
      if (early_blank) {
        if ((rc = flush(qb, S2))) return rc;
        ev_bl_next = c->ev_blank2[round & 1];
        HIPCHK(c, hipEventRecord(ev_bl_next, S2));
      }
      if (Bnext > 0) {
        on(S1);
        if (round >= 1) HIPCHK(c, hipStreamWaitEvent(S1, c->ev_sumsq[(round + 1) & 1], 0));
        if ((rc = lrh_fft1_b(c, 0, p->timf1p_px, p->fft1_pa, Bnext))) return rc;
        advance_fft1(c, p, Bnext);
        if (!fuse) HIPCHK(c, hipEventRecord(c->ev_fft1, S1));
      }
      if (have_prev && (rc = main_tail())) return rc;
      ev_bl_tail = ev_bl_next;
      c->rec = &qt; rc = round_tail(c, p); c->rec = nullptr;
      if (rc) return rc;
      have_prev = true;
      left -= B; B = Bnext; round++;
    }
    if (carry_ok) {                                        // the last round stays parked: the next call (or flush_pending) issues it
      c->pend_b.swap(qb); c->pend_t.swap(qt);
      c->pend = true; c->pend_tail_flushed = tail_flushed; c->pend_batch = batch;
      clv_guard.ok = true; return LRH_OK;
    }
    if ((rc = side_blanker())) return rc;
    if (c->clv_wait && (rc = clever_late_finish(c, p))) return rc;     // linear blanker: the last round's resume point, then its dumb blanker
    if ((rc = main_tail())) return rc;
    HIPCHK(c, hipEventRecord(c->ev_side, S2)); HIPCHK(c, hipStreamWaitEvent(S1, c->ev_side, 0));
    clv_guard.ok = true; return LRH_OK;
  }
  // the side stream starts after everything already queued on the main stream
  HIPCHK(c, hipEventRecord(c->ev_side, S1)); HIPCHK(c, hipStreamWaitEvent(S2, c->ev_side, 0));
  int left = nblocks, round = 0;
  int B = left < batch ? left : batch;
  // prologue: fft1(0) on the main stream, its sums on the side stream
  on(S1); if ((rc = lrh_fft1_b(c, 0, p->timf1p_px, p->fft1_pa, B))) return rc;
  advance_fft1(c, p, B);
  HIPCHK(c, hipEventRecord(c->ev_fft1, S1));
  on(S2); HIPCHK(c, hipStreamWaitEvent(S2, c->ev_fft1, 0));
  if ((rc = sums(B))) return rc;
  if (!fuse) HIPCHK(c, hipEventRecord(c->ev_sumsq[round & 1], S2));
  while (left > 0) {
    const int Bnext = (left - B) < batch ? (left - B) : batch;       // size of round k+1 (0 at the end)
    const int lim_cnt = p->fft1_liminfo_cnt, lim_pa = p->fft1_sumsq_pa, lim_ctr = p->fft1_sumsq_counter;   // fft1_c's counters after round k's sums
    // main: timf2(k)
    on(S1); if ((rc = lrh_make_timf2(c, p, B))) return rc;
    HIPCHK(c, hipEventRecord(c->ev_timf2, S1));
    // side: blanker(k)
    on(S2); HIPCHK(c, hipStreamWaitEvent(S2, c->ev_timf2, 0));
    if (fuse) { if ((rc = sums_follow(S2))) return rc; HIPCHK(c, hipEventRecord(c->ev_sumsq[round & 1], S2)); }
    if ((rc = lrh_first_noise_blanker(c, p))) return rc;
    HIPCHK(c, hipEventRecord(c->ev_blank, S2));
    if ((rc = limiter1(lim_cnt, lim_pa, lim_ctr))) return rc;          // behind the blanker: fft2(k) on the main stream waits for that one
    if (Bnext > 0) {
      // main: fft1(k+1) once the sums of round k-1 (which read the ring slots it overwrites) are done
      on(S1);
      if (round >= 1) HIPCHK(c, hipStreamWaitEvent(S1, c->ev_sumsq[(round + 1) & 1], 0));
      if ((rc = lrh_fft1_b(c, 0, p->timf1p_px, p->fft1_pa, Bnext))) return rc;
      advance_fft1(c, p, Bnext);
      HIPCHK(c, hipEventRecord(c->ev_fft1, S1));
      // side: sumsq/slowsum(k+1)
      on(S2); HIPCHK(c, hipStreamWaitEvent(S2, c->ev_fft1, 0));
      if ((rc = sums(Bnext))) return rc;
      if (!fuse) HIPCHK(c, hipEventRecord(c->ev_sumsq[(round + 1) & 1], S2));
    }
    // main: fft2(k) + mix1(k) after the blanker released the data; their power sums / waterfall go to the side stream
    on(S1); HIPCHK(c, hipStreamWaitEvent(S1, c->ev_blank, 0));
    c->split_fft2_tail = true;
    rc = round_tail(c, p);
    c->split_fft2_tail = false; c->last_main_ev = nullptr;
    if (rc) return rc;
    if ((rc = limiter2())) return rc;
    left -= B; B = Bnext; round++;
  }
  // join: later API calls are ordered on the main stream only
  HIPCHK(c, hipEventRecord(c->ev_side, S2)); HIPCHK(c, hipStreamWaitEvent(S1, c->ev_side, 0));
  clv_guard.ok = true; return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- outputs
static int export_impl(lrh_ctx *c, lrh_ring ring, void *dst, size_t off, size_t cnt, hipMemcpyKind kind, bool wait = true, int *ticket = nullptr);
static int export_collect(lrh_ctx *c, int slot);
int lrh_export_device_async(lrh_ctx *c, lrh_ring ring, void *dst, size_t off, size_t cnt) try { return export_impl(c, ring, dst, off, cnt, hipMemcpyDeviceToDevice, false); } LRH_CATCH(c)
void *lrh_stream(lrh_ctx *c) { return c ? (void *)c->stream : nullptr; }
int lrh_export(lrh_ctx *c, lrh_ring ring, void *dst, size_t off, size_t cnt) try { return export_impl(c, ring, dst, off, cnt, hipMemcpyDeviceToHost); } LRH_CATCH(c)
int lrh_export_device(lrh_ctx *c, lrh_ring ring, void *dst, size_t off, size_t cnt) try { return export_impl(c, ring, dst, off, cnt, hipMemcpyDeviceToDevice); } LRH_CATCH(c)
static int export_impl(lrh_ctx *c, lrh_ring ring, void *dst, size_t off, size_t cnt, hipMemcpyKind kind, bool wait, int *ticket)
{
  if (ticket) *ticket = 0;
  if (!c || !dst) return LRH_EINVAL;
  LRH_ENTER(c);
  // transforms of the fft1_b workers (their own streams) and a parked transform: only a reader of what fft1 itself leaves needs them --
  // everything behind fft1_c / make_timf2 was enqueued by calls that have joined them already
  // (the sums of fft1_c are behind transforms that call has issued itself: a reader of them does not have to issue what the workers have noted since)
  if (ring == LRH_RING_TIMF1 || ring == LRH_RING_FFT1_FLOAT || c->f1_have) { const int rc_ = join_handles(c); if (rc_) return rc_; }   // (incl. the workers' noted blocks)
  const void *src; size_t esz = 4, total;
  switch (ring) {
    case LRH_RING_TIMF1: src = c->d_timf1; esz = 2; total = c->cfg.timf1_bytes / 2; break;
    case LRH_RING_FFT1_FLOAT: src = c->d_fft1; total = (size_t)c->cfg.max_fft1n * 2 * c->N1; break;
    case LRH_RING_FFT1_SUMSQ: src = c->d_sumsq; total = c->cfg.fft1_sumsq_bufsize; break;
    case LRH_RING_FFT1_SLOWSUM: src = c->d_slowsum; total = c->N1; break;
    case LRH_RING_TIMF2_FLOAT: {
      // the device keeps weak and strong planar; rebuild the reference layout {wRe,wIm,sRe,sIm} per sample
      total = 4 * (size_t)c->cfg.timf2pow_size;
      if (off + cnt > total || (off & 3) || (cnt & 3)) return LRH_EINVAL;
      const size_t s0 = off / 4, ns = cnt / 4;
      if (ns == 0) return LRH_OK;
      if (kind == hipMemcpyDeviceToHost && c->h_stage) {   // through the staging buffer: weak samples in its first half, strong in the second, interleaved on the way out
        const size_t per = LRH_STAGE_BYTES / 16;
        float *out = (float *)dst;
        for (size_t at = 0; at < ns; at += per) {
          const size_t n = ns - at < per ? ns - at : per;
          const float2 *hw = (const float2 *)c->h_stage, *hs = hw + per;
          HIPCHK(c, hipMemcpyAsync(c->h_stage, c->d_timf2w + s0 + at, n * 8, hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipMemcpyAsync(c->h_stage + per * 8, c->d_timf2s + s0 + at, n * 8, hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream));
          for (size_t i = 0; i < n; i++) { float *o = out + 4 * (at + i); o[0] = hw[i].x; o[1] = hw[i].y; o[2] = hs[i].x; o[3] = hs[i].y; }
        }
        return LRH_OK;
      }
      HIPCHK(c, hipMemcpy2DAsync(dst, 16, c->d_timf2w + s0, 8, 8, ns, kind, c->stream));
      HIPCHK(c, hipMemcpy2DAsync((char *)dst + 8, 16, c->d_timf2s + s0, 8, 8, ns, kind, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      return LRH_OK;
    }
    case LRH_RING_TIMF2_PWR: src = c->d_pwr; total = c->cfg.timf2pow_size; break;
    case LRH_RING_FFT2_FLOAT: src = c->d_fft2; total = (size_t)c->cfg.max_fft2n * 2 * c->N2; break;
    case LRH_RING_FFT2_POWER:
      src = c->d_power2; total = (size_t)c->cfg.max_fft2n * c->N2;
      if (c->fft2_fused) {                               // the hot path keeps only the sums: |X|^2 of the requested span on demand
        if (off > total || cnt > total - off) return LRH_EINVAL;
        LRH_WRITES(c, RB(LRH_RING_FFT2_POWER));
        HIPCHK(c, launch_power_of(c->d_fft2 + off, c->d_power2 + off, cnt, c->stream));
      }
      break;
    case LRH_RING_FFT2_POWERSUM: src = c->d_powersum2; total = c->N2; break;
    case LRH_RING_WG_WATERF: src = c->d_waterf; esz = 2; total = (size_t)c->cfg.wf_lines * c->cfg.wf_xpixels; break;
    case LRH_RING_TIMF3_FLOAT: src = c->d_timf3; total = c->cfg.timf3_size; break;
    case LRH_RING_TIMF2_BLOCKPOWER: src = c->d_blockpower; total = c->cfg.timf2_blockpower_size; break;
    case LRH_RING_FFT3: src = c->d_fft3; total = (size_t)c->cfg.max_fft3n * 2 * c->N3; break;
    case LRH_RING_BASEB_RAW: src = c->d_baseb; total = 2 * (size_t)c->cfg.baseband_size; break;
    case LRH_RING_FFT2_XYPOWER: if (!c->d_xypower) return fail(c, LRH_ESTATE, "blanker_channels != 2"); src = c->d_xypower; total = (size_t)c->cfg.max_fft2n * 4 * c->N2; break;
    case LRH_RING_FFT2_XYSUM: if (!c->d_xysum) return fail(c, LRH_ESTATE, "blanker_channels != 2"); src = c->d_xysum; total = 4 * (size_t)c->N2; break;
    case LRH_RING_FFT1_CORRSUM: if (!c->corr_on) return fail(c, LRH_ESTATE, "lrh_set_correlation first"); src = c->d_corrsum; total = 2 * (size_t)c->cfg.fft1_sumsq_bufsize; break;
    case LRH_RING_FFT1_SLOWCORR: if (!c->corr_on) return fail(c, LRH_ESTATE, "lrh_set_correlation first"); src = c->d_slowcorr; total = 2 * (size_t)c->N1; break;
    case LRH_RING_FFT1_SLOWCORR_TOT: if (!c->corr_on) return fail(c, LRH_ESTATE, "lrh_set_correlation first"); src = c->d_slowcorr_tot; total = 2 * (size_t)c->N1; esz = 8; break;
    default: return LRH_EINVAL;
  }
  if (off + cnt > total) return LRH_EINVAL;
  bool direct = false;                                     // destination page-locked by the caller (lrh_host_register): the copy engine writes it itself, no staging slot, no memcpy, no size limit
  if (kind == hipMemcpyDeviceToHost) for (const auto &r : c->host_regs) if ((char *)dst >= r.first && (char *)dst + cnt * esz <= r.first + r.second) { direct = true; break; }
  if (kind == hipMemcpyDeviceToHost && wait && (direct || cnt * esz <= LRH_OUT_SLOT_BYTES) && cnt > 0 && !c->in_dsp && !c->rec && c->out_ok) {
    // page-locked slot, copy stream, wait outside the lock (see lrh_ctx::stream_out)
    int slot = -1;
    for (int i = 0; i < LRH_NOUT; i++) if (!c->out_busy[i]) { slot = i; break; }
    if (slot >= 0) {
      if (!c->stream_out) HIPCHK(c, hipStreamCreateWithFlags(&c->stream_out, hipStreamNonBlocking));
      if (!c->ev_out_src[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_out_src[slot], hipEventDisableTiming));
      if (!c->ev_out_done[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_out_done[slot], hipEventDisableTiming));
      if (!direct && !c->h_out[slot] && lrh_host_malloc(&c->h_out[slot], LRH_OUT_SLOT_BYTES) != hipSuccess) { c->h_out[slot] = nullptr; return fail(c, LRH_ENOMEM, "lrh_host_malloc(read-back slot)"); }
      c->out_busy[slot] = true; c->out_ring[slot] = (int)ring;   // (a stage that rewrites this ring queues behind the copy: order_behind_readbacks)
      hipError_t e_ = hipEventRecord(c->ev_out_src[slot], c->stream);
      if (e_ == hipSuccess) e_ = hipStreamWaitEvent(c->stream_out, c->ev_out_src[slot], 0);
      if (e_ == hipSuccess) e_ = hipMemcpyAsync(direct ? dst : c->h_out[slot], (const char *)src + off * esz, cnt * esz, hipMemcpyDeviceToHost, c->stream_out);
      if (e_ == hipSuccess) e_ = hipEventRecord(c->ev_out_done[slot], c->stream_out);
      if (e_ != hipSuccess) { c->out_busy[slot] = false; return fail(c, LRH_EDEVICE, "read-back", e_); }
      c->out_dst[slot] = direct ? nullptr : dst; c->out_bytes[slot] = cnt * esz;
      if (ticket) { *ticket = slot + 1; return LRH_OK; }     // lrh_export_begin: the caller collects it with lrh_export_end
      lk_.unlock();
      return export_collect(c, slot);
    }
  }
  if (ticket) *ticket = 0;                                  // no slot (or a span beyond a slot's size): done here, nothing to collect
  if (kind == hipMemcpyDeviceToHost) { HIPCHK(c, stage_d2h(c, dst, (const char *)src + off * esz, cnt * esz, c->stream)); return LRH_OK; }   // (a host destination is always waited for)
  HIPCHK(c, hipMemcpyAsync(dst, (const char *)src + off * esz, cnt * esz, kind, c->stream));
  if (wait) HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
// second half of a read-back: called WITHOUT the context's lock
static int export_collect(lrh_ctx *c, int slot)
{
  const hipError_t e_ = hipEventSynchronize(c->ev_out_done[slot]);
  if (e_ == hipSuccess && c->out_dst[slot]) memcpy(c->out_dst[slot], c->h_out[slot], c->out_bytes[slot]);
  { std::lock_guard<std::recursive_mutex> lk(c->mtx); c->out_busy[slot] = false; }
  if (e_ != hipSuccess) return fail(c, LRH_EDEVICE, "hipEventSynchronize(read-back)", e_);
  return LRH_OK;
}
int lrh_export_begin(lrh_ctx *c, lrh_ring ring, void *dst, size_t off, size_t cnt, int *ticket)
try {
  if (!ticket) return LRH_EINVAL;
  return export_impl(c, ring, dst, off, cnt, hipMemcpyDeviceToHost, true, ticket);
}
LRH_CATCH(c)
int lrh_export_end(lrh_ctx *c, int ticket)
try {
  if (!c || ticket < 0 || ticket > LRH_NOUT) return LRH_EINVAL;
  if (ticket == 0) return LRH_OK;
  if (!c->out_busy[ticket - 1]) return fail(c, LRH_ESTATE, "lrh_export_end: no read-back under this ticket");
  return export_collect(c, ticket - 1);
}
LRH_CATCH(c)

int lrh_stage_wait(lrh_ctx *c, int stage) { return lrh_stage_wait_lag(c, stage, 0); }
int lrh_stage_wait_lag(lrh_ctx *c, int stage, int lag)
try {
  if (!c || stage < 0 || stage >= LRH_STAGE_COUNT || lag < 0 || lag > 3) return LRH_EINVAL;
  hipEvent_t ev = nullptr;
  { LRH_LOCK(c);
    if (c->stage_lag_env >= 0) lag = c->stage_lag_env;
    if (lag == 0) { if (c->stage_valid[stage]) ev = stage == LRH_STAGE_TIMF2 ? c->ev_timf2_done : c->ev_stage[stage]; }
    else if (c->stage_seq[stage] > (unsigned)lag) ev = c->ev_stage_ring[stage][(c->stage_seq[stage] - 1 - lag) & 3]; }   // the call `lag` before the newest (none yet: nothing to wait for)
  // the event belongs to the context for its life; the thread that waits is the one that makes this stage's calls, so it is not re-recorded meanwhile
  if (ev && hipEventSynchronize(ev) != hipSuccess) return fail(c, LRH_EDEVICE, "hipEventSynchronize(stage)");
  return LRH_OK;
}
LRH_CATCH(c)
static int stage_mark(lrh_ctx *c, int stage)          // behind the device work a stage call has just enqueued on c->cur (stage calls only, not inside lrh_wideband_dsp)
{
  if (c->in_dsp || c->rec) return LRH_OK;
  if (!c->ev_stage[stage]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage[stage], hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_stage[stage], c->cur));
  c->stage_valid[stage] = true;
  return stage_ring_mark(c, stage);
}
static int stage_ring_mark(lrh_ctx *c, int stage)     // the same point of the stream in the history lrh_stage_wait_lag reads (one more record: ~3 us)
{
  hipEvent_t &ev = c->ev_stage_ring[stage][c->stage_seq[stage] & 3];
  if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(ev, c->cur));
  c->stage_seq[stage]++;
  return LRH_OK;
}

int lrh_export_timf2_net(lrh_ctx *c, float *dst, int timf2_pt, int count, float gain, float strong)
try {
  LRH_ENTER(c);
  if (!c || !dst || count < 0 || count > c->cfg.timf2pow_size || (timf2_pt & 3)) return LRH_EINVAL;
  if (!count) return LRH_OK;
  if ((size_t)count > c->net_cap) {
    if (c->d_net) lrh_dev_free(c->d_net);
    c->d_net = nullptr; c->net_cap = 0;
    if (lrh_dev_malloc((void **)&c->d_net, (size_t)count * sizeof(float2)) != hipSuccess) return fail(c, LRH_ENOMEM, "lrh_dev_malloc(timf2 net staging)");
    c->net_cap = count;
  }
  HIPCHK(c, launch_timf2_net(c->d_timf2w, c->d_timf2s, c->timf2pow_mask, (timf2_pt & c->timf2_mask) / 4, count, gain, strong, c->d_net, c->stream));
  HIPCHK(c, stage_d2h(c, dst, c->d_net, (size_t)count * sizeof(float2), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)
// NET_RXOUT_FFT1 (wcw.c:1024-1043, network.c:383-388): the transform as fft1_b leaves it, i.e. before fft1_c's filter correction.
// The hot path applies that correction in k_fft1's store, so the ring never holds the bare transform; a sender of this stage
// gets it recomputed from the timf1 ring into a staging buffer (the same kernels with a unit filter table; the mirror-image
// step and the direction flip, which belong to fft1_b, included) -- one extra fft1 pass, only for installations that multicast
// this stage.
int lrh_export_fft1_net(lrh_ctx *c, float *dst, int timf1p_ref, int batch)
try {
  LRH_ENTER(c);
  if (!c || !dst || batch < 1 || batch > c->cfg.max_batch) return LRH_EINVAL;
  if (c->rec) return fail(c, LRH_ESTATE, "not inside lrh_wideband_dsp");
  int cap = 1; while (cap < batch) cap <<= 1;
  if (cap > c->fft1net_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_fft1net) lrh_dev_free(c->d_fft1net);
    c->d_fft1net = nullptr; c->fft1net_cap = 0;
    if (lrh_dev_malloc((void **)&c->d_fft1net, (size_t)cap * c->N1 * sizeof(float2)) != hipSuccess) return fail(c, LRH_ENOMEM, "lrh_dev_malloc(fft1 net staging)");
    c->fft1net_cap = cap;
  }
  if (!c->d_unitcorr) {
    if (lrh_dev_malloc((void **)&c->d_unitcorr, sizeof(float2) * c->N1) != hipSuccess) return fail(c, LRH_ENOMEM, "lrh_dev_malloc(unit filter table)");
    std::vector<float2> one(c->N1, make_float2(1.f, 0.f));
    HIPCHK(c, stage_h2d(c, c->d_unitcorr, one.data(), sizeof(float2) * c->N1, c->stream));
  }
  if (c->in_pending) { const int rc_ = wait_for_input(c, c->stream); if (rc_) return rc_; }
  Fft1Args a; a.spare_cus = 0;
  const int C = c->cfg.timf1_frame_channels > 1 ? c->cfg.timf1_frame_channels : 1;
  const int esz = c->cfg.timf1_dword_input ? 8 : 4;
  a.timf1 = c->d_timf1; a.ring_mask = c->cfg.timf1_bytes / esz - 1; a.dword = c->cfg.timf1_dword_input != 0;
  a.shift_i = c->cfg.sample_shift > 0 ? -c->cfg.sample_shift : 0; a.shift_q = c->cfg.sample_shift < 0 ? c->cfg.sample_shift : 0;
  a.chan_count = C; a.chan_index = C > 1 ? c->cfg.timf1_channel_index : 0;
  a.p0_first = ((timf1p_ref & c->timf1_bytemask) / (esz * C) - c->I1) & (a.ring_mask / C);
  a.step = c->M1; a.window = c->d_window1; a.filtercorr = c->d_unitcorr; a.tw = c->d_tw1; a.out = c->d_fft1net;
  a.first_nb = 0; a.nb_mask = c->fft1net_cap - 1; a.direction = c->cfg.fft1_direction; a.xcd = 0; a.batch = batch;
  a.real = c->cfg.timf1_real_input != 0; a.stamps = nullptr;
  if (c->d_foldcorr || a.real) a.direction = 1;
  if (c->fft1_big) {                                       // four-step fft1 through the scratch of handle 0 (this stream's)
    const size_t need = (size_t)batch * c->N1;
    if (c->fft1_scratch_cap[0] < need) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (c->d_fft1_scratch[0]) lrh_dev_free(c->d_fft1_scratch[0]);
      c->d_fft1_scratch[0] = nullptr; c->fft1_scratch_cap[0] = 0;
      const int rc_ = dev_alloc(c, &c->d_fft1_scratch[0], need, false); if (rc_) return rc_;
      c->fft1_scratch_cap[0] = need;
    }
    Fft1BigArgs g; g.f = a; g.tw_a = c->d_tw1a; g.tw_b = c->d_tw1b; g.tw_big = c->d_tw1; g.scratch = c->d_fft1_scratch[0];
    HIPCHK(c, launch_fft1_big(c->cfg.fft1_n, g, batch, c->stream));
  } else
  HIPCHK(c, launch_fft1(c->cfg.fft1_n, a, batch, c->stream));
  if (a.real) {
    RealSplitArgs r; r.spec = c->d_fft1net; r.first_nb = 0; r.nb_mask = a.nb_mask; r.n = c->N1; r.filtercorr = c->d_unitcorr; r.direction = c->cfg.fft1_direction;
    HIPCHK(c, launch_realsplit(r, batch, c->stream));
  } else if (c->d_foldcorr) {
    FoldcorrArgs f; f.spec = c->d_fft1net; f.first_nb = 0; f.nb_mask = a.nb_mask; f.n = c->N1; f.foldcorr = c->d_foldcorr; f.filtercorr = c->d_unitcorr; f.direction = c->cfg.fft1_direction;
    HIPCHK(c, launch_foldcorr(f, batch, c->stream));
  }
  HIPCHK(c, stage_d2h(c, dst, c->d_fft1net, (size_t)batch * c->N1 * sizeof(float2), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRH_OK;
}
LRH_CATCH(c)

// Issue what lrh_wideband_dsp still holds back from its last round (the one-round-late schedule kept across calls) without waiting for it:
// afterwards everything the calls so far have produced is ordered on the context's stream, which is what a consumer chained on that stream
// (lrh_stream: a collective, a torch ExternalStream) needs.  Every entry point that reads or changes results does this by itself.
int lrh_flush(lrh_ctx *c)
try {
  LRH_ENTER(c);
  return c ? LRH_OK : LRH_EINVAL;
}
LRH_CATCH(c)

int lrh_sync(lrh_ctx *c)
try {
  if (!c) return LRH_EINVAL;
  LRH_ENTER(c);
  { const int rc_ = join_handles(c); if (rc_) return rc_; }  // blocks the fft1_b workers have noted
  // every stream of the context: producer copies (the header lets the caller reuse `src` after this) and table uploads too
  { std::lock_guard<std::mutex> lk_in(c->mtx_in); const int rc_ = flush_input_locked(c); if (rc_) return rc_; }
  for (hipStream_t s : { c->stream_in, c->stream3, c->stream, c->stream2, c->stream_nb, c->stream_sel }) if (s) HIPCHK(c, hipStreamSynchronize(s));
  for (int h = 1; h < LRH_MAX_HANDLES; h++) if (c->hstream[h]) HIPCHK(c, hipStreamSynchronize(c->hstream[h]));
  return sellim_install(c, c->sel_seq);
}
LRH_CATCH(c)

int lrh_timer_start(lrh_ctx *c) try { LRH_ENTER(c); if (!c) return LRH_EINVAL; HIPCHK(c, hipEventRecord(c->t0, c->stream)); return LRH_OK; } LRH_CATCH(c)
int lrh_timer_stop(lrh_ctx *c, float *ms)
try {
  LRH_ENTER(c);
  if (!c || !ms) return LRH_EINVAL;
  HIPCHK(c, hipEventRecord(c->t1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->t1));
  HIPCHK(c, hipEventElapsedTime(ms, c->t0, c->t1));
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_profile_enable(lrh_ctx *c, int on)
try {
  LRH_ENTER(c);
  if (!c) return LRH_EINVAL;
  prof_collect(c); c->prof = on != 0; c->prof_keep_schedule = on == 2; c->prof_tot.clear();
  return LRH_OK;
}
LRH_CATCH(c)
int lrh_profile_get(lrh_ctx *c, const char *kernel, double *total_ms, long *launches)
try {
  LRH_ENTER(c);
  if (!c || !kernel) return LRH_EINVAL;
  if (!strcmp(kernel, "host:mix1_phases")) { if (total_ms) *total_ms = c->host_ms_phases; if (launches) *launches = c->host_n_phases; return LRH_OK; }
  if (!strcmp(kernel, "host:staging_wait")) { if (total_ms) *total_ms = c->host_ms_wait; if (launches) *launches = c->host_n_dsp; return LRH_OK; }
  if (!strcmp(kernel, "host:wideband_dsp_cpu")) { if (total_ms) *total_ms = c->host_cpu_ms_dsp; if (launches) *launches = c->host_n_dsp; return LRH_OK; }
  if (!strcmp(kernel, "host:wideband_dsp")) { if (total_ms) *total_ms = c->host_ms_dsp; if (launches) *launches = c->host_n_dsp; return LRH_OK; }
  prof_collect(c);
  auto it = c->prof_tot.find(kernel);
  if (total_ms) *total_ms = it == c->prof_tot.end() ? 0 : it->second.ms;
  if (launches) *launches = it == c->prof_tot.end() ? 0 : it->second.n;
  return LRH_OK;
}
LRH_CATCH(c)

// ---------------------------------------------------------------------------------------------- test signal
static inline uint64_t splitmix64(uint64_t &x) { uint64_t z = (x += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
struct Xo { uint64_t s[4]; uint64_t next() { uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
            double uni() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); } };

void lrh_synth_defaults(lrh_synth *s, int fft1_size, int channel)
{
  memset(s, 0, sizeof *s);
  s->seed = 0x4C494E52ULL + (uint64_t)channel; s->noise_sigma = 64.0f; s->fft_size = fft1_size;
  const float bins[8] = {-6000, -3111, -517, 37, 1024, 2999.5f, 4500, 7000};
  const float amps[8] = {8000, 50, 200, 1000, 100, 300, 20, 4000};
  s->ncarriers = 8;
  for (int i = 0; i < 8; i++) { s->carrier_bin[i] = bins[i] * (float)fft1_size / 16384.0f; s->carrier_amp[i] = amps[i]; }
  s->pulse_period = 9973; s->pulse_len = 3; s->pulse_amp = 20000.0f; s->chan_phase = 0.7f * channel;
}

// Position-addressable: any (first_sample, nsamples) window of the same infinite sequence gives the same bytes.
int lrh_synth_iq(const lrh_synth *s, int64_t first, int64_t n, int16_t *dst)
try {
  if (!s || !dst || n < 0 || s->ncarriers < 0 || s->ncarriers > 16 || s->fft_size <= 0) return LRH_EINVAL;
  const int64_t BLK = 4096;
  for (int64_t pos = first; pos < first + n;) {
    const int64_t blk = pos >= 0 ? pos / BLK : -((-pos + BLK - 1) / BLK);
    const int64_t b0 = blk * BLK;
    int64_t e = b0 + BLK; if (e > first + n) e = first + n;
    uint64_t sm = s->seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(blk + 0x100000));
    Xo g; for (int i = 0; i < 4; i++) g.s[i] = splitmix64(sm);
    double cr[16], ci[16], wr[16], wi[16];
    for (int k = 0; k < s->ncarriers; k++) {
      const double cyc = (double)s->carrier_bin[k] / s->fft_size;
      double ph = cyc * (double)b0; ph -= floor(ph);   // phase at the block start, in cycles
      const double a = 2 * PI_L * ph + s->chan_phase;
      cr[k] = s->carrier_amp[k] * cos(a); ci[k] = s->carrier_amp[k] * sin(a);
      wr[k] = cos(2 * PI_L * cyc); wi[k] = sin(2 * PI_L * cyc);
    }
    for (int64_t i = b0; i < e; i++) {
      // Box-Muller on the block's own stream: every sample of the block is drawn whether or not it is requested
      const double u1 = g.uni(), u2 = g.uni();
      const double r = s->noise_sigma * sqrt(-2.0 * log(u1));
      double re = r * cos(2 * PI_L * u2), im = r * sin(2 * PI_L * u2);
      for (int k = 0; k < s->ncarriers; k++) {
        re += cr[k]; im += ci[k];
        const double t = cr[k] * wr[k] - ci[k] * wi[k]; ci[k] = cr[k] * wi[k] + ci[k] * wr[k]; cr[k] = t;
      }
      if (s->pulse_period > 0 && i >= 0) {
        const int64_t pi = i / s->pulse_period, off = i - pi * s->pulse_period;
        if (off < s->pulse_len) {
          uint64_t h = s->seed ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(pi + 1)); const double pa = 2 * PI_L * ((splitmix64(h) >> 11) * (1.0 / 9007199254740992.0));
          re += s->pulse_amp * cos(pa); im += s->pulse_amp * sin(pa);
        }
      }
      if (i >= pos) {
        double a = nearbyint(re), b = nearbyint(im);
        if (a > 32767) a = 32767; if (a < -32767) a = -32767; if (b > 32767) b = 32767; if (b < -32767) b = -32767;
        dst[2 * (i - first)] = (int16_t)a; dst[2 * (i - first) + 1] = (int16_t)b;
      }
    }
    pos = e;
  }
  return LRH_OK;
}
LRH_CATCH_NOCTX

}  // extern "C"
