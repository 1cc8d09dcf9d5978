// hip_emu_rt.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h): the emulated HIP runtime -- devices, device-tagged allocations
// behind page protection, streams and events with device identity -- and a fake RCCL.  None of this is linked into the product.
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <set>
#include <string>

#include "hip_emu.h"
#include "rccl_emu.h"

namespace emu {
Stats stats;
bool fault_no_device_guard = false;
bool fault_drop_waits = false;
// allocation-failure injection: p2hot_emu_fault("fail_malloc_at", k) -- the k-th hipMalloc / hipHostMalloc from now AND every later
// one return hipErrorOutOfMemory until disarmed (the device is full: a retry fails too); "fail_malloc_once", k -- only the k-th
// (a transient failure: what the block cache's "give the cached blocks back and try once more" recovers from).  `malloc_calls`
// counts the calls since the last arming so that a sweep knows how far to go
std::atomic<long long> fault_malloc_countdown{0}, malloc_calls{0};
std::atomic<bool> fault_malloc_persistent{false}, fault_malloc_tripped{false};
static bool malloc_fault_fires() {
    ++malloc_calls;
    if (fault_malloc_tripped.load()) return true;
    long long c = fault_malloc_countdown.load();
    while (c > 0) {
        if (fault_malloc_countdown.compare_exchange_weak(c, c - 1)) {
            if (c == 1 && fault_malloc_persistent.load()) fault_malloc_tripped = true;
            return c == 1;
        }
    }
    return false;
}

namespace {
struct Alloc {
    size_t bytes = 0, mapped = 0;  // mapped: the accessible pages; ONE more page behind them is a guard that is never accessible
    uintptr_t user = 0;            // what hipMalloc returned: the allocation ENDS (to 16 bytes) where the guard page begins
    int device = -1;               // -1: host (pinned) memory
    bool open = true;              // pages currently readable / writable
};
std::mutex mu;                        // registry lock; never held while user memory is touched
std::map<uintptr_t, Alloc> allocs;    // base address -> allocation
int n_devices = 0;
int open_device = 0;                  // the device whose allocations are currently accessible (process-wide)
thread_local int tls_device = 0;
thread_local std::string tls_error;
thread_local hipError_t tls_sticky = hipSuccess;
std::set<std::pair<int, int>> peer_enabled;

// ---- streams are QUEUES (P2HOT_EMU_ASYNC=0 switches back to immediate execution).  A launch, an asynchronous copy, a memset, an
// event record, a stream wait and a collective are appended to their stream and run only when somebody NEEDS the result: a
// hipStreamSynchronize / hipEventSynchronize / hipDeviceSynchronize, a synchronous copy, a hipFree -- or another stream's wait on
// an event recorded behind them.  Work nobody waits for stays undone, so a consumer that forgot its hipStreamWaitEvent runs BEFORE
// its producer and reads stale memory, and a host read of a result before the synchronisation reads stale memory: the missing
// dependency of an asynchronous program shows as a wrong answer in the CPU tier.  (One adversarial schedule -- as late as legal --
// not all of them.)  The copy rules follow the runtime's documentation: from pageable host memory the source is captured when the
// call returns; into pageable host memory the call is synchronous; pinned and device memory are read and written in stream order.
struct EventState {
    int device;
    unsigned long long enqueued = 0, done = 0;  // hipEventRecord calls so far / records that have executed
};
struct Stream;
struct Coll;
struct Task {
    enum Kind { RUN, WAIT, RECORD, COLL } kind = RUN;
    std::function<void()> fn;
    std::shared_ptr<EventState> ev;
    unsigned long long ver = 0;
    Stream *src = nullptr;  // WAIT: the stream the awaited record sits on
    std::shared_ptr<Coll> coll;
};
struct Stream {
    int device;
    std::deque<Task> q;
    bool busy = false;  // being advanced further up the call stack (blocked on a wait): reaching it again is a circular wait
};
struct Event {
    int device;
    std::shared_ptr<EventState> st;
    Stream *rec = nullptr;  // where the latest record was enqueued
};
std::set<void *> live_streams, live_events;
std::recursive_mutex sched_mu;        // the scheduler: held while queued work executes (kernels are emulated one at a time)
std::map<int, Stream *> null_streams;  // the default stream of every device

size_t page() {
    static const size_t p = (size_t)sysconf(_SC_PAGESIZE);
    return p;
}

int device_count() {
    if (n_devices == 0) {
        const char *e = getenv("P2HOT_EMU_DEVICES");
        n_devices = e ? std::max(1, atoi(e)) : 8;
    }
    return n_devices;
}

hipError_t violation(hipError_t code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tls_error = std::string("hip_emu: ") + buf;
    ++stats.violations;
    if (getenv("P2HOT_EMU_TRACE")) fprintf(stderr, "%s\n", tls_error.c_str());
    return code;
}

std::map<uintptr_t, Alloc>::iterator find_alloc(const void *p) {  // caller holds mu
    const uintptr_t a = (uintptr_t)p;
    auto it = allocs.upper_bound(a);
    if (it == allocs.begin()) return allocs.end();
    --it;
    return a < it->first + std::max(it->second.bytes, it->second.mapped + (it->second.device >= 0 ? page() : 0)) ? it : allocs.end();
}

void set_open(std::map<uintptr_t, Alloc>::iterator it, bool open) {  // caller holds mu
    if (it->second.device < 0 || it->second.open == open) return;
    mprotect((void *)it->first, it->second.mapped, open ? (PROT_READ | PROT_WRITE) : PROT_NONE);
    it->second.open = open;
}

// makes the device allocations containing the given pointers accessible for the lifetime of the object, whatever device is
// current (the runtime's own copies: peer copies, device <-> host copies issued from another device's context)
struct Access {
    std::vector<uintptr_t> reopened;
    Access(std::initializer_list<const void *> ptrs) {
        std::lock_guard<std::mutex> l(mu);
        for (const void *p : ptrs) {
            auto it = find_alloc(p);
            if (it != allocs.end() && !it->second.open) {
                set_open(it, true);
                reopened.push_back(it->first);
            }
        }
    }
    ~Access() {
        std::lock_guard<std::mutex> l(mu);
        for (uintptr_t b : reopened) {
            auto it = allocs.find(b);
            if (it != allocs.end() && it->second.device != open_device) set_open(it, false);
        }
    }
};

struct sigaction old_segv;
void on_segv(int sig, siginfo_t *info, void *uctx) {
    const void *addr = info->si_addr;
    bool ours = false, let_through = false;
    int dev = -1;
    size_t bytes = 0;
    uintptr_t base = 0;
    bool overrun = false;
    if (mu.try_lock()) {
        auto it = find_alloc(addr);
        if (it != allocs.end() && it->second.device >= 0 && (uintptr_t)addr >= it->first + it->second.mapped) {
            overrun = true;
            dev = it->second.device;
            bytes = it->second.bytes;
            base = it->second.user;
        } else if (it != allocs.end() && it->second.device >= 0 && !it->second.open) {
            ours = true;
            dev = it->second.device;
            bytes = it->second.bytes;
            base = it->first;
            if (dev == tls_device) {  // another thread switched the process-wide window: this thread's own device owns the pages
                set_open(it, true);
                let_through = true;
            }
        }
        mu.unlock();
    }
    if (let_through) return;
    if (overrun) {
        fprintf(stderr,
                "hip_emu: DEVICE MEMORY OVERRUN: address %p is %zu bytes past the end of a %zu-byte allocation of device %d (hipMalloc returned %p) "
                "-- an index ran off the end of a device buffer\n",
                addr, (size_t)((uintptr_t)addr - (base + bytes)), bytes, dev, (void *)base);
        fflush(stderr);
        abort();
    }
    if (ours) {
        fprintf(stderr,
                "hip_emu: DEVICE MEMORY FAULT: address %p lies in a %zu-byte allocation of device %d (base %p) but device %d is current "
                "-- a kernel, a host loop or a copy touched another GPU's memory without an explicit peer copy\n",
                addr, bytes, dev, (void *)base, tls_device);
        fflush(stderr);
        abort();
    }
    // not an emulated device allocation: hand the fault to whoever was installed before (pytest's faulthandler, the default action)
    sigaction(SIGSEGV, &old_segv, nullptr);
    (void)sig, (void)uctx;
}

void install_handler() {
    static bool done = false;
    if (done) return;
    done = true;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv);
}

hipError_t check_stream(hipStream_t s, const char *what) {
    if (!s) return hipSuccess;  // the null stream is the current device's
    std::lock_guard<std::mutex> l(mu);
    if (!live_streams.count(s)) return violation(hipErrorInvalidResourceHandle, "%s: the stream handle %p was destroyed or never created", what, s);
    const int d = ((Stream *)s)->device;
    if (d != tls_device)
        return violation(hipErrorInvalidResourceHandle, "%s on a stream of device %d while device %d is current (hipSetDevice is missing or stale)", what, d,
                         tls_device);
    return hipSuccess;
}

int stream_device(hipStream_t s) { return s ? ((Stream *)s)->device : tls_device; }
}  // namespace

int current_device() { return tls_device; }

int device_of(const void *p) {
    std::lock_guard<std::mutex> l(mu);
    auto it = find_alloc(p);
    return it == allocs.end() ? -1 : it->second.device;
}

namespace {
// P2HOT_EMU_ASYNC: unset / "1" = as late as legal; "0" = immediate execution (the old model); "random:<seed>" = as late as legal
// plus, at every enqueue and before every synchronisation, a random number of steps on randomly chosen streams -- other LEGAL
// schedules between the two extremes (every step honours its stream's order and its waits), e.g. an overwrite running before a
// lagging reader that nothing orders it against
bool async_mode() {
    static const bool on = !(getenv("P2HOT_EMU_ASYNC") && !strcmp(getenv("P2HOT_EMU_ASYNC"), "0"));
    return on;
}
std::mt19937_64 *chaos_rng() {
    static std::mt19937_64 *rng = []() -> std::mt19937_64 * {
        const char *e = getenv("P2HOT_EMU_ASYNC");
        return e && !strncmp(e, "random:", 7) ? new std::mt19937_64(strtoull(e + 7, nullptr, 10)) : nullptr;
    }();
    return rng;
}
Stream *resolve(hipStream_t s) {  // caller holds sched_mu
    if (s) return (Stream *)s;
    auto it = null_streams.find(tls_device);
    if (it == null_streams.end()) it = null_streams.emplace(tls_device, new Stream{tls_device}).first;
    return it->second;
}

// queued work of device d runs with d current and its pages open, whatever the calling thread's device is
struct ExecScope {
    int saved;
    static void switch_to(int d) {
        tls_device = d;
        std::lock_guard<std::mutex> l(mu);
        if (d != open_device) {
            for (auto it = allocs.begin(); it != allocs.end(); ++it)
                if (it->second.device >= 0) set_open(it, it->second.device == d);
            open_device = d;
        }
    }
    explicit ExecScope(int d) : saved(tls_device) {
        if (d != saved) switch_to(d);
    }
    ~ExecScope() {
        if (tls_device != saved) switch_to(saved);
    }
};

struct Coll {  // one collective of a communicator whose ranks live in this process: executes when EVERY rank's stream has reached it
    int kind;  // 0 broadcast, 1 all-gather
    size_t bytes;
    int root;
    struct Part {
        Stream *st;
        int rank;
        const void *send;
        void *recv;
    };
    std::vector<Part> parts;
};
void run_collective(const Coll &c);  // (fake RCCL section)

hipError_t deadlock(const char *what, Stream *s) {
    return violation((hipError_t)999, "%s: circular wait -- a stream of device %d waits (directly or through other streams) for work queued behind its own wait: "
                                      "the real runtime hangs here", what, s->device);
}

hipError_t step(Stream *s);
hipError_t drive_event(Stream *src, EventState *ev, unsigned long long ver, const char *what) {
    while (ev->done < ver) {
        if (!src || src->busy) return deadlock(what, src ? src : resolve(nullptr));
        if (src->q.empty()) return violation((hipError_t)999, "%s: the awaited event record is not queued anywhere (the recording stream was reset?)", what);
        hipError_t rc = step(src);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}
hipError_t step(Stream *s) {  // executes the head of s (caller holds sched_mu, s->q is not empty)
    Task &t = s->q.front();
    static const bool trace = getenv("P2HOT_EMU_TRACE_SCHED") != nullptr;
    if (trace) fprintf(stderr, "hip_emu: stream %p (device %d) runs %s, %zu queued behind it\n", (void *)s, s->device,
                       t.kind == Task::RUN ? "work" : t.kind == Task::WAIT ? "a wait" : t.kind == Task::RECORD ? "an event record" : "a collective", s->q.size() - 1);
    hipError_t rc = hipSuccess;
    switch (t.kind) {
        case Task::WAIT: {
            s->busy = true;
            std::shared_ptr<EventState> ev = t.ev;
            rc = drive_event(t.src, ev.get(), t.ver, "hipStreamWaitEvent");
            s->busy = false;
            if (rc == hipSuccess) s->q.pop_front();
            return rc;
        }
        case Task::RECORD:
            t.ev->done = std::max(t.ev->done, t.ver);
            s->q.pop_front();
            return hipSuccess;
        case Task::COLL: {
            std::shared_ptr<Coll> c = t.coll;
            s->busy = true;
            for (auto &p : c->parts) {
                while (rc == hipSuccess && !(!p.st->q.empty() && p.st->q.front().kind == Task::COLL && p.st->q.front().coll == c)) {
                    if (p.st->busy || p.st->q.empty()) {
                        rc = deadlock("collective", p.st);
                        break;
                    }
                    rc = step(p.st);
                }
                if (rc != hipSuccess) break;
            }
            s->busy = false;
            if (rc != hipSuccess) return rc;
            run_collective(*c);
            for (auto &p : c->parts) p.st->q.pop_front();
            return hipSuccess;
        }
        default: {
            std::function<void()> fn = std::move(t.fn);
            s->q.pop_front();
            s->busy = true;
            {
                ExecScope scope(s->device);
                fn();
            }
            s->busy = false;
            return hipSuccess;
        }
    }
}
void chaos() {  // caller holds sched_mu
    std::mt19937_64 *rng = chaos_rng();
    if (!rng) return;
    static bool inside = false;
    if (inside) return;
    inside = true;
    for (unsigned n = (unsigned)((*rng)() % 4); n > 0; --n) {
        std::vector<Stream *> ready;
        {
            std::lock_guard<std::mutex> l(mu);
            for (void *p : live_streams)
                if (!((Stream *)p)->q.empty() && !((Stream *)p)->busy) ready.push_back((Stream *)p);
        }
        for (auto &kv : null_streams)
            if (!kv.second->q.empty() && !kv.second->busy) ready.push_back(kv.second);
        if (ready.empty()) break;
        (void)step(ready[(size_t)((*rng)() % ready.size())]);  // (an error here resurfaces at the synchronisation that needs the work)
    }
    inside = false;
}
hipError_t drain(Stream *s) {
    chaos();
    while (!s->q.empty()) {
        if (s->busy) return deadlock("synchronisation", s);
        hipError_t rc = step(s);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}
hipError_t drain_device(int d) {
    std::vector<Stream *> all;
    {
        std::lock_guard<std::mutex> l(mu);
        for (void *p : live_streams)
            if (((Stream *)p)->device == d) all.push_back((Stream *)p);
    }
    auto it = null_streams.find(d);
    if (it != null_streams.end()) all.push_back(it->second);
    for (Stream *st : all) {
        hipError_t rc = drain(st);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}
// appends work to a stream, or runs it now in the immediate mode
hipError_t enqueue(hipStream_t stream, std::function<void()> fn) {
    if (!async_mode()) {
        fn();
        return hipSuccess;
    }
    std::lock_guard<std::recursive_mutex> g(sched_mu);
    Task t;
    t.kind = Task::RUN;
    t.fn = std::move(fn);
    resolve(stream)->q.push_back(std::move(t));
    chaos();
    return hipSuccess;
}
enum MemKind { PAGEABLE, PINNED, DEVICE };
MemKind mem_kind(const void *p) {
    std::lock_guard<std::mutex> l(mu);
    auto it = find_alloc(p);
    return it == allocs.end() ? PAGEABLE : it->second.device < 0 ? PINNED : DEVICE;
}
}  // namespace

void launch_on(hipStream_t stream, dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
    hipError_t e = check_stream(stream, "kernel launch");
    if (e != hipSuccess) {
        tls_sticky = e;  // reported by the hipGetLastError that follows every launch
        return;
    }
    (void)enqueue(stream, [grid, block, shmem, body]() { launch(grid, block, shmem, body); });
}
}  // namespace emu

using namespace emu;

hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= device_count()) return violation(hipErrorInvalidDevice, "hipSetDevice(%d): the emulated node has %d devices", d, device_count());
    tls_device = d;
    std::lock_guard<std::mutex> l(mu);
    if (d != open_device) {
        ++stats.device_switches;
        for (auto it = allocs.begin(); it != allocs.end(); ++it)
            if (it->second.device >= 0) set_open(it, it->second.device == d);
        open_device = d;
    }
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
    *d = tls_device;
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int *n) {
    *n = device_count();
    return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned) {
    if (peer < 0 || peer >= device_count() || peer == tls_device) return violation(hipErrorInvalidDevice, "hipDeviceEnablePeerAccess(%d) from device %d", peer, tls_device);
    std::lock_guard<std::mutex> l(mu);
    peer_enabled.insert({tls_device, peer});
    return hipSuccess;
}

hipError_t hipMalloc(void **p, size_t n) {
    install_handler();
    if (malloc_fault_fires()) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    // the allocation is placed so that it ENDS (rounded up to 16 bytes) at a guard page: an index that runs off the end of a device
    // buffer faults on the spot, in every kernel, copy and host loop of the tier (what AddressSanitizer did while device memory was malloc'ed)
    const size_t want = ((n ? n : 1) + 15) / 16 * 16;
    const size_t mapped = (want + page() - 1) / page() * page();
    void *m = mmap(nullptr, mapped + page(), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    mprotect((char *)m + mapped, page(), PROT_NONE);
    std::lock_guard<std::mutex> l(mu);
    Alloc a;
    a.bytes = n;
    a.mapped = mapped;
    a.user = (uintptr_t)m + (mapped - want);
    a.device = tls_device;
    a.open = true;
    auto it = allocs.emplace((uintptr_t)m, a).first;
    if (tls_device != open_device) set_open(it, false);  // (another thread's device holds the window; this thread's first touch reopens it)
    *p = (void *)a.user;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    if (async_mode()) {  // hipFree waits for the device's outstanding work (it may still use the block)
        const int d = device_of(p);
        std::lock_guard<std::recursive_mutex> g(sched_mu);
        hipError_t rc = drain_device(d >= 0 ? d : tls_device);
        if (rc != hipSuccess) return rc;
    }
    std::lock_guard<std::mutex> l(mu);
    auto it = find_alloc(p);
    if (it == allocs.end() || it->second.device < 0 || it->second.user != (uintptr_t)p)
        return violation(hipErrorInvalidValue, "hipFree(%p): not the base of a live device allocation", p);
    munmap((void *)it->first, it->second.mapped + page());
    allocs.erase(it);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    if (malloc_fault_fires()) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> l(mu);
    Alloc a;
    a.bytes = n ? n : 1;
    a.device = -1;
    allocs.emplace((uintptr_t)*p, a);
    return hipSuccess;
}
hipError_t hipHostFree(void *p) {
    if (!p) return hipSuccess;
    if (async_mode()) {
        std::lock_guard<std::recursive_mutex> g(sched_mu);
        hipError_t rc = drain_device(tls_device);
        if (rc != hipSuccess) return rc;
    }
    {
        std::lock_guard<std::mutex> l(mu);
        auto it = allocs.find((uintptr_t)p);
        if (it == allocs.end() || it->second.device >= 0) return violation(hipErrorInvalidValue, "hipHostFree(%p): not a live pinned allocation", p);
        allocs.erase(it);
    }
    free(p);
    return hipSuccess;
}

static void count_peer(const void *d, const void *s, size_t n) {
    const int dd = device_of(d), ds = device_of(s);
    if (dd >= 0 && ds >= 0 && dd != ds) {
        ++stats.peer_copies;
        stats.peer_bytes += n;
    }
}
hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t stream) {
    hipError_t e = check_stream(stream, "hipMemcpyAsync");
    if (e != hipSuccess) return e;
    if (width == 0 || height == 0) return hipSuccess;
    count_peer(d, s, width * height);
    auto copy = [=](const void *from, size_t from_pitch) {
        Access acc{d, from};
        for (size_t r = 0; r < height; ++r) memmove((char *)d + r * dpitch, (const char *)from + r * from_pitch, width);
    };
    if (!async_mode()) {
        copy(s, spitch);
        return hipSuccess;
    }
    const MemKind dk = mem_kind(d), sk = mem_kind(s);
    if (dk == PAGEABLE) {  // "the function will return only once the copy has completed": everything queued before it runs now
        std::lock_guard<std::recursive_mutex> g(sched_mu);
        hipError_t rc = drain(resolve(stream));
        if (rc != hipSuccess) return rc;
        copy(s, spitch);
        return hipSuccess;
    }
    if (sk == PAGEABLE) {  // the source is staged before the call returns (the caller may reuse it); the device side is stream ordered
        auto staged = std::make_shared<std::vector<unsigned char>>(width * height);
        for (size_t r = 0; r < height; ++r) memcpy(staged->data() + r * width, (const char *)s + r * spitch, width);
        return enqueue(stream, [=]() {
            (void)staged;
            copy(staged->data(), width);
        });
    }
    return enqueue(stream, [=]() { copy(s, spitch); });
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t stream) {
    return hipMemcpy2DAsync(d, n, s, n, n, n ? 1 : 0, k, stream);
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) {
    hipError_t rc = hipMemcpyAsync(d, s, n, k, nullptr);
    return rc != hipSuccess ? rc : hipStreamSynchronize(nullptr);
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t stream) {
    hipError_t e = check_stream(stream, "hipMemsetAsync");
    if (e != hipSuccess) return e;
    const int dd = device_of(d);
    if (dd >= 0 && dd != tls_device) return violation(hipErrorInvalidValue, "hipMemsetAsync on memory of device %d while device %d is current", dd, tls_device);
    return enqueue(stream, [=]() { memset(d, v, n); });
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    Stream *st = new Stream{tls_device};
    std::lock_guard<std::mutex> l(mu);
    live_streams.insert(st);
    *s = st;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    std::lock_guard<std::recursive_mutex> g(sched_mu);
    {
        std::lock_guard<std::mutex> l(mu);
        if (!s || !live_streams.count(s)) return violation(hipErrorInvalidResourceHandle, "hipStreamDestroy(%p): not a live stream", s);
    }
    hipError_t rc = drain((Stream *)s);  // the queued work still completes
    std::lock_guard<std::mutex> l(mu);
    live_streams.erase(s);
    delete (Stream *)s;
    return rc;
}
hipError_t hipStreamSynchronize(hipStream_t s) {  // any device's stream may be synchronised from any thread
    std::lock_guard<std::recursive_mutex> g(sched_mu);
    if (s) {
        std::lock_guard<std::mutex> l(mu);
        if (!live_streams.count(s)) return violation(hipErrorInvalidResourceHandle, "hipStreamSynchronize(%p): the stream was destroyed or never created", s);
    }
    return drain(resolve(s));
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
    Event *ev = new Event{tls_device, std::make_shared<EventState>()};
    ev->st->device = tls_device;
    std::lock_guard<std::mutex> l(mu);
    live_events.insert(ev);
    *e = ev;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
    std::lock_guard<std::mutex> l(mu);
    if (!e || !live_events.erase(e)) return violation(hipErrorInvalidResourceHandle, "hipEventDestroy(%p): not a live event", e);
    delete (Event *)e;  // queued records / waits keep the shared state alive: destroying an event with pending work is legal
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    hipError_t rc = check_stream(s, "hipEventRecord");
    if (rc != hipSuccess) return rc;
    std::lock_guard<std::recursive_mutex> g(sched_mu);
    std::lock_guard<std::mutex> l(mu);
    if (!live_events.count(e)) return violation(hipErrorInvalidResourceHandle, "hipEventRecord: the event %p was destroyed or never created", e);
    Event *ev = (Event *)e;
    if (ev->device != stream_device(s))
        return violation(hipErrorInvalidResourceHandle, "hipEventRecord: an event of device %d recorded on a stream of device %d", ev->device, stream_device(s));
    const unsigned long long ver = ++ev->st->enqueued;
    if (!async_mode()) {
        ev->st->done = ver;
        return hipSuccess;
    }
    Task t;
    t.kind = Task::RECORD;
    t.ev = ev->st;
    t.ver = ver;
    ev->rec = resolve(s);
    ev->rec->q.push_back(std::move(t));
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    hipError_t rc = check_stream(s, "hipStreamWaitEvent");
    if (rc != hipSuccess) return rc;
    std::lock_guard<std::recursive_mutex> g(sched_mu);
    std::lock_guard<std::mutex> l(mu);
    if (!live_events.count(e)) return violation(hipErrorInvalidResourceHandle, "hipStreamWaitEvent: the event %p was destroyed or never created", e);
    // (an event of ANOTHER device is fine: that is how the ranks' streams are ordered against each other)
    Event *ev = (Event *)e;
    if (fault_drop_waits) return hipSuccess;  // test hook: every stream dependency of the program is "forgotten"
    if (ev->st->enqueued == 0 || ev->st->done >= ev->st->enqueued) return hipSuccess;  // never recorded / already complete: no wait
    Task t;
    t.kind = Task::WAIT;
    t.ev = ev->st;
    t.ver = ev->st->enqueued;  // the record the event holds NOW; a later re-record does not move this wait
    t.src = ev->rec;
    resolve(s)->q.push_back(std::move(t));
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    std::lock_guard<std::recursive_mutex> g(sched_mu);
    Event *ev = (Event *)e;
    {
        std::lock_guard<std::mutex> l(mu);
        if (!live_events.count(e)) return violation(hipErrorInvalidResourceHandle, "hipEventSynchronize: the event %p was destroyed or never created", e);
    }
    return drive_event(ev->rec, ev->st.get(), ev->st->enqueued, "hipEventSynchronize");
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    std::lock_guard<std::recursive_mutex> g(sched_mu);
    std::lock_guard<std::mutex> l(mu);
    if (!live_events.count(a) || !live_events.count(b)) return violation(hipErrorInvalidResourceHandle, "hipEventElapsedTime: dead event");
    for (hipEvent_t e : {a, b})
        if (((Event *)e)->st->done < ((Event *)e)->st->enqueued)
            return violation((hipError_t)600, "hipEventElapsedTime: an event has not completed yet (hipErrorNotReady): synchronise first");
    *ms = 0.f;  // the emulator has no clock
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() {
    std::lock_guard<std::recursive_mutex> g(sched_mu);
    return drain_device(tls_device);
}
hipError_t hipGetLastError() {
    hipError_t e = tls_sticky;
    tls_sticky = hipSuccess;
    return e;
}
const char *hipGetErrorString(hipError_t e) {
    if (e != hipSuccess && !tls_error.empty()) return tls_error.c_str();
    return e == hipErrorOutOfMemory ? "hip_emu: out of memory" : "hip_emu: error";
}
hipError_t hipMemGetInfo(size_t *f, size_t *t) {
    *f = *t = (size_t)1 << 34;
    return hipSuccess;
}

// ================================================================= fake RCCL
// What the multi-GPU layer needs from the collectives library, with the rules whose violation hangs or corrupts on the real
// one turned into errors: a communicator rank is bound to ONE device; the stream of a collective must belong to it; the
// buffers must live on it (or be untracked test memory); one thread driving several ranks must bracket their calls with
// ncclGroupStart / ncclGroupEnd, and every rank of the communicator must post the same sequence of collectives.
namespace {
struct Clique;
struct Shared;
struct Comm {
    int rank, nranks, device;
    std::shared_ptr<Clique> clique;
    Shared *shm = nullptr;  // a communicator whose ranks are PROCESSES (ncclCommInitRank with nranks > 1): see below
    std::string shm_name;
    unsigned seq = 0;
};
struct Clique {
    std::vector<Comm *> members;
};
struct Op {
    int kind;  // 0 broadcast, 1 all-gather
    Comm *comm;
    const void *send;
    void *recv;
    size_t bytes;
    int root;
    void *stream;
};
Stream *stream_of(void *stream, int device) {  // caller holds sched_mu
    if (stream) return (Stream *)stream;
    auto it = null_streams.find(device);
    if (it == null_streams.end()) it = null_streams.emplace(device, new Stream{device}).first;
    return it->second;
}
std::set<void *> live_comms;
thread_local int group_depth = 0;
thread_local std::vector<Op> queued;
thread_local std::string nccl_error;
unsigned long long uid_counter = 0;

int nccl_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    nccl_error = std::string("rccl_emu: ") + buf;
    ++stats.violations;
    if (getenv("P2HOT_EMU_TRACE")) fprintf(stderr, "%s\n", nccl_error.c_str());
    return code;
}
size_t dtype_bytes(int dt) { return dt == 0 || dt == 1 ? 1 : dt == 2 || dt == 3 || dt == 7 ? 4 : dt == 4 || dt == 5 || dt == 8 ? 8 : dt == 6 ? 2 : 0; }

void move(void *d, const void *s, size_t n) {
    if (d == s || n == 0) return;
    Access acc{d, s};
    memmove(d, s, n);
    stats.nccl_bytes += n;
}

}  // namespace
namespace emu {
namespace {
void run_collective(const Coll &c) {
    if (c.kind == 0) {
        const void *src = nullptr;
        for (auto &p : c.parts)
            if (p.rank == c.root) src = p.send;
        for (auto &p : c.parts) move(p.recv, src, c.bytes);
        ++stats.nccl_broadcasts;
    } else {
        // every rank's contribution is staged first: an in-place all-gather reads a slot another rank's copy may overwrite
        std::vector<std::vector<unsigned char>> stage(c.parts.size());
        for (auto &p : c.parts) {
            stage[(size_t)p.rank].resize(c.bytes);
            Access acc{p.send};
            memcpy(stage[(size_t)p.rank].data(), p.send, c.bytes);
        }
        for (auto &p : c.parts)
            for (size_t q = 0; q < stage.size(); ++q) {
                Access acc{p.recv};
                memcpy((char *)p.recv + q * c.bytes, stage[q].data(), c.bytes);
                stats.nccl_bytes += c.bytes;
            }
        ++stats.nccl_allgathers;
    }
}
}  // namespace
}  // namespace emu
namespace {

int validate(const char *what, Comm *c, const void *send, void *recv, hipStream_t stream) {
    {
        std::lock_guard<std::mutex> l(mu);
        if (!live_comms.count(c)) return nccl_fail(4, "%s: the communicator %p was destroyed or never created", what, (void *)c);
        if (stream && !live_streams.count(stream)) return nccl_fail(4, "%s: rank %d was handed a dead stream", what, c->rank);
    }
    const int sd = stream ? ((Stream *)stream)->device : tls_device;
    if (sd != c->device)
        return nccl_fail(5, "%s: rank %d of the communicator lives on device %d but its stream belongs to device %d (ncclInvalidUsage)", what, c->rank, c->device, sd);
    for (const void *p : {send, (const void *)recv}) {
        const int d = device_of(p);
        if (d >= 0 && d != c->device)
            return nccl_fail(5, "%s: rank %d (device %d) was handed a buffer of device %d", what, c->rank, c->device, d);
    }
    return 0;
}

// ---- one process per rank (bench.py --gpus N, plonky2_amd.distributed.Communicator(transport="rccl")): the ranks meet in a POSIX
// shared-memory segment named by the unique id.  Every collective is announced (sequence number, kind, bytes, root) and checked
// against every other rank's announcement before a byte moves; data travels through a staging area in pieces.  A rank that
// never arrives -- a rank that posted fewer collectives, took another branch, or died -- is a TIMEOUT with a message, where
// the real library hangs.  Device ids are node-global: two ranks on one device are refused as RCCL refuses them.
constexpr size_t kStagePerRank = 1 << 20;
constexpr int kMaxRanks = 16;
struct Announce {
    unsigned seq, kind, root;
    unsigned long long bytes;
};
struct Shared {
    std::atomic<unsigned> magic, arrived, generation, dead, joined;
    int nranks;
    int device[kMaxRanks];
    Announce ann[kMaxRanks];
    alignas(64) unsigned char stage[1];  // nranks * kStagePerRank
};
constexpr unsigned kMagic = 0x70326874;

long timeout_ms() {
    const char *e = getenv("P2HOT_EMU_RCCL_TIMEOUT_MS");
    return e ? atol(e) : 60000;
}

// sense-reversing barrier over the segment; 0 = everybody arrived, else an error code with nccl_error set
int shared_barrier(Comm *c, const char *what) {
    Shared *sh = c->shm;
    if (sh->dead.load()) return nccl_fail(6, "%s: another rank of the communicator already failed (rank %d gives up instead of waiting for ever)", what, c->rank);
    const unsigned gen = sh->generation.load();
    if (sh->arrived.fetch_add(1) + 1 == (unsigned)sh->nranks) {
        sh->arrived.store(0);
        sh->generation.fetch_add(1);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const long limit = timeout_ms();
    while (sh->generation.load() == gen) {
        if (sh->dead.load()) return nccl_fail(6, "%s: another rank of the communicator failed while rank %d waited for it", what, c->rank);
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > limit) {
            sh->dead.store(1);
            return nccl_fail(6, "%s: rank %d waited %ld ms for the other ranks of the communicator: they posted fewer collectives, took another "
                                "path or died -- the real library hangs here", what, c->rank, limit);
        }
        usleep(50);
    }
    return 0;
}

int run_shared(Comm *c, std::vector<Op> &ops) {
    Shared *sh = c->shm;
    const size_t R = (size_t)c->nranks;
    for (auto &o : ops) {
        if (async_mode()) {  // a collective between PROCESSES runs when it is posted (the other ranks' hosts are not ours to schedule):
            std::lock_guard<std::recursive_mutex> g(sched_mu);  // everything queued in front of it on its stream runs first
            if (drain(stream_of(o.stream, c->device)) != hipSuccess) return nccl_fail(6, "a collective's stream could not be advanced: %s", tls_error.c_str());
        }
        sh->ann[c->rank] = Announce{++c->seq, (unsigned)o.kind, (unsigned)o.root, (unsigned long long)o.bytes};
        int rc = shared_barrier(c, o.kind ? "ncclAllGather" : "ncclBroadcast");
        if (rc) return rc;
        const Announce a0 = sh->ann[0];
        int bad = -1;
        for (size_t r = 0; r < R; ++r)
            if (sh->ann[r].seq != a0.seq || sh->ann[r].kind != a0.kind || sh->ann[r].root != a0.root || sh->ann[r].bytes != a0.bytes) bad = (int)r;
        if (a0.kind == 0 && a0.root >= R) bad = 0;
        if (bad >= 0) {  // every rank sees the same table and fails the same way
            const Announce b = sh->ann[bad];
            rc = nccl_fail(5, "collective %u differs between ranks (rank 0: kind %u, %llu bytes, root %u; rank %d: #%u kind %u, %llu bytes, root %u): mismatched "
                              "collectives corrupt or hang on the real library", a0.seq, a0.kind, a0.bytes, a0.root, bad, b.seq, b.kind, b.bytes, b.root);
            (void)shared_barrier(c, "mismatch");
            return rc;
        }
        for (size_t off = 0; off < o.bytes || off == 0; off += kStagePerRank) {
            const size_t len = std::min(kStagePerRank, o.bytes - off);
            if (o.kind == 0) {
                if (c->rank == o.root && len) {
                    Access acc{o.send};
                    memcpy(sh->stage, (const char *)o.send + off, len);
                }
                if ((rc = shared_barrier(c, "ncclBroadcast"))) return rc;
                if (len && !(c->rank == o.root && o.send == o.recv)) {
                    Access acc{o.recv};
                    memcpy((char *)o.recv + off, sh->stage, len);
                }
            } else {
                if (len) {
                    Access acc{o.send};
                    memcpy(sh->stage + (size_t)c->rank * kStagePerRank, (const char *)o.send + off, len);
                }
                if ((rc = shared_barrier(c, "ncclAllGather"))) return rc;
                for (size_t q = 0; q < R && len; ++q) {
                    Access acc{o.recv};
                    memcpy((char *)o.recv + q * o.bytes + off, sh->stage + q * kStagePerRank, len);
                }
            }
            stats.nccl_bytes += len;
            if ((rc = shared_barrier(c, "collective")))  return rc;  // the staging area is free again
            if (o.bytes == 0) break;
        }
        ++(o.kind ? stats.nccl_allgathers : stats.nccl_broadcasts);
    }
    return 0;
}

int run_group(std::vector<Op> &ops) {
    {  // communicators whose ranks are processes: this process posts for ONE rank of each; in posting order
        std::map<Comm *, std::vector<Op>> shared;
        std::vector<Comm *> order;
        std::vector<Op> local;
        for (auto &o : ops) {
            if (!o.comm->shm) {
                local.push_back(o);
                continue;
            }
            if (!shared.count(o.comm)) order.push_back(o.comm);
            shared[o.comm].push_back(o);
        }
        for (Comm *c : order) {
            int rc = run_shared(c, shared[c]);
            if (rc) return rc;
        }
        ops.swap(local);
        if (ops.empty()) return 0;
    }
    std::map<Clique *, std::map<int, std::vector<Op>>> by;  // clique -> rank -> ops in posting order
    for (auto &o : ops) by[o.comm->clique.get()][o.comm->rank].push_back(o);
    for (auto &kv : by) {
        Clique *cl = kv.first;
        auto &ranks = kv.second;
        const size_t n_ops = ranks.begin()->second.size();
        if (ranks.size() != cl->members.size())
            return nccl_fail(5, "a collective group covers %zu of the %zu ranks of its communicator: the others never arrive (would hang)", ranks.size(), cl->members.size());
        for (auto &r : ranks)
            if (r.second.size() != n_ops) return nccl_fail(5, "rank %d posted %zu collectives, rank %d posted %zu (would hang)", r.first, r.second.size(), ranks.begin()->first, n_ops);
        for (size_t k = 0; k < n_ops; ++k) {
            const Op &o0 = ranks.begin()->second[k];
            for (auto &r : ranks) {
                const Op &o = r.second[k];
                if (o.kind != o0.kind || o.bytes != o0.bytes || o.root != o0.root)
                    return nccl_fail(5, "collective %zu differs between ranks (kind %d/%d, bytes %zu/%zu, root %d/%d): mismatched collectives", k, o.kind, o0.kind, o.bytes, o0.bytes, o.root, o0.root);
            }
            if (o0.kind == 0 && (o0.root < 0 || o0.root >= (int)cl->members.size())) return nccl_fail(4, "ncclBroadcast: root %d out of range", o0.root);
            auto c = std::make_shared<Coll>();
            c->kind = o0.kind, c->bytes = o0.bytes, c->root = o0.root;
            std::lock_guard<std::recursive_mutex> g(sched_mu);
            for (auto &r : ranks) c->parts.push_back(Coll::Part{stream_of(r.second[k].stream, r.second[k].comm->device), r.first, r.second[k].send, r.second[k].recv});
            if (!async_mode()) {
                run_collective(*c);
                continue;
            }
            for (auto &p : c->parts) {  // it executes when every rank's stream has reached it
                Task t;
                t.kind = Task::COLL;
                t.coll = c;
                p.st->q.push_back(std::move(t));
            }
        }
    }
    return 0;
}

int post(Op o) {
    if (group_depth > 0) {
        queued.push_back(o);
        return 0;
    }
    if (o.comm->nranks > 1 && !o.comm->shm)
        return nccl_fail(5, "a collective on a %d-rank communicator outside ncclGroupStart / ncclGroupEnd from the one thread that drives every rank: "
                            "the real library blocks here for ever", o.comm->nranks);
    std::vector<Op> one{o};
    return run_group(one);
}
}  // namespace

extern "C" {
int emu_ncclGetUniqueId(char *id128) {
    memset(id128, 0, 128);
    const unsigned long long c = ++uid_counter;
    snprintf(id128, 128, "/p2hot_emu_rccl_%d_%llu_%llx", (int)getpid(), c,
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return 0;
}
int emu_ncclCommInitRank(void **comm, int nranks, const char *id128, int rank) {
    if (nranks < 1 || rank < 0 || rank >= nranks) return nccl_fail(4, "ncclCommInitRank: rank %d of %d", rank, nranks);
    if (nranks == 1) {
        Comm *c = new Comm{0, 1, tls_device, std::make_shared<Clique>()};
        c->clique->members.push_back(c);
        std::lock_guard<std::mutex> l(mu);
        live_comms.insert(c);
        *comm = c;
        return 0;
    }
    if (nranks > kMaxRanks) return nccl_fail(4, "ncclCommInitRank: the emulated node has at most %d ranks", kMaxRanks);
    if (!id128 || id128[0] != '/' || memchr(id128, 0, 128) == nullptr)
        return nccl_fail(4, "ncclCommInitRank: rank %d was handed a unique id that no ncclGetUniqueId produced (was rank 0's id distributed?)", rank);
    const size_t bytes = sizeof(Shared) + (size_t)nranks * kStagePerRank;
    int fd = shm_open(id128, O_CREAT | O_EXCL | O_RDWR, 0600);
    const bool creator = fd >= 0;
    if (!creator) fd = shm_open(id128, O_RDWR, 0600);
    if (fd < 0) return nccl_fail(2, "ncclCommInitRank: shm_open(%s): %s", id128, strerror(errno));
    if (creator && ftruncate(fd, (off_t)bytes) != 0) {
        close(fd);
        shm_unlink(id128);
        return nccl_fail(2, "ncclCommInitRank: ftruncate: %s", strerror(errno));
    }
    if (!creator) {  // the creator sizes the segment before anybody may map it
        struct stat st;
        const auto t0 = std::chrono::steady_clock::now();
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < bytes) {
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) {
                close(fd);
                return nccl_fail(6, "ncclCommInitRank: rank %d found the segment of another world size (ranks disagree on nranks?)", rank);
            }
            usleep(50);
        }
    }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return nccl_fail(2, "ncclCommInitRank: mmap: %s", strerror(errno));
    Shared *sh = (Shared *)m;
    if (creator) {
        sh->nranks = nranks;
        for (int r = 0; r < kMaxRanks; ++r) sh->device[r] = -1;
        sh->magic.store(kMagic);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (sh->magic.load() != kMagic) {
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) {
                munmap(m, bytes);
                return nccl_fail(6, "ncclCommInitRank: rank %d never saw the segment initialised", rank);
            }
            usleep(50);
        }
    }
    Comm *c = new Comm{rank, nranks, tls_device, std::make_shared<Clique>()};
    c->shm = sh;
    c->shm_name = id128;
    auto bail = [&](int rc) {
        munmap(m, bytes);
        if (rank == 0) shm_unlink(id128);
        delete c;
        return rc;
    };
    if (sh->nranks != nranks) {
        sh->dead.store(1);
        return bail(nccl_fail(5, "ncclCommInitRank: rank %d says %d ranks, the communicator has %d", rank, nranks, sh->nranks));
    }
    if (sh->device[rank] != -1) {
        sh->dead.store(1);
        return bail(nccl_fail(5, "ncclCommInitRank: two processes claim rank %d", rank));
    }
    sh->device[rank] = tls_device;
    int rc = shared_barrier(c, "ncclCommInitRank");  // the real call returns when every rank has joined
    if (rc) return bail(rc);
    for (int r = 0; r < nranks; ++r)
        for (int q = 0; q < r; ++q)
            if (sh->device[r] == sh->device[q]) {
                rc = nccl_fail(5, "ncclCommInitRank: Duplicate GPU detected: rank %d and rank %d both on device %d", q, r, sh->device[r]);
                (void)shared_barrier(c, "ncclCommInitRank");
                return bail(rc);
            }
    if ((rc = shared_barrier(c, "ncclCommInitRank"))) return bail(rc);
    if (rank == 0) shm_unlink(id128);  // every rank has mapped it: the name can go, the memory lives until the last unmap
    std::lock_guard<std::mutex> l(mu);
    live_comms.insert(c);
    *comm = c;
    return 0;
}
int emu_ncclCommInitAll(void **comms, int ndev, const int *devlist) {
    if (ndev < 1) return nccl_fail(4, "ncclCommInitAll: %d devices", ndev);
    for (int i = 0; i < ndev; ++i) {
        const int d = devlist ? devlist[i] : i;
        if (d < 0 || d >= device_count()) return nccl_fail(4, "ncclCommInitAll: device %d does not exist", d);
        for (int j = 0; j < i; ++j)
            if ((devlist ? devlist[j] : j) == d) return nccl_fail(4, "ncclCommInitAll: Duplicate GPU detected: rank %d and rank %d both on device %d", j, i, d);
    }
    auto cl = std::make_shared<Clique>();
    std::lock_guard<std::mutex> l(mu);
    for (int i = 0; i < ndev; ++i) {
        Comm *c = new Comm{i, ndev, devlist ? devlist[i] : i, cl};
        cl->members.push_back(c);
        live_comms.insert(c);
        comms[i] = c;
    }
    return 0;
}
int emu_ncclCommDestroy(void *comm) {
    std::lock_guard<std::mutex> l(mu);
    if (!live_comms.erase(comm)) return nccl_fail(4, "ncclCommDestroy(%p): not a live communicator", comm);
    Comm *c = (Comm *)comm;
    if (c->shm) munmap(c->shm, sizeof(Shared) + (size_t)c->nranks * kStagePerRank);
    delete c;
    return 0;
}
int emu_ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, void *stream) {
    Comm *c = (Comm *)comm;
    int rc = validate("ncclBroadcast", c, send, recv, stream);
    if (rc) return rc;
    if (!dtype_bytes(dtype)) return nccl_fail(4, "ncclBroadcast: unknown datatype %d", dtype);
    return post(Op{0, c, send, recv, count * dtype_bytes(dtype), root, stream});
}
int emu_ncclAllGather(const void *send, void *recv, size_t sendcount, int dtype, void *comm, void *stream) {
    Comm *c = (Comm *)comm;
    int rc = validate("ncclAllGather", c, send, recv, stream);
    if (rc) return rc;
    if (!dtype_bytes(dtype)) return nccl_fail(4, "ncclAllGather: unknown datatype %d", dtype);
    return post(Op{1, c, send, recv, sendcount * dtype_bytes(dtype), 0, stream});
}
int emu_ncclGroupStart() {
    ++group_depth;
    return 0;
}
int emu_ncclGroupEnd() {
    if (group_depth == 0) return nccl_fail(5, "ncclGroupEnd without ncclGroupStart");
    if (--group_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(queued);
    return run_group(ops);
}
const char *emu_ncclGetErrorString(int rc) { return rc == 0 ? "no error" : (nccl_error.empty() ? "rccl_emu: error" : nccl_error.c_str()); }

// ---- test hooks exported by the emulator build only
void p2hot_emu_stats(unsigned long long out[8]) {
    out[0] = stats.peer_copies, out[1] = stats.peer_bytes, out[2] = stats.nccl_broadcasts, out[3] = stats.nccl_allgathers;
    out[4] = stats.nccl_bytes, out[5] = stats.device_switches, out[6] = stats.violations, out[7] = (unsigned long long)tls_device;
}
int p2hot_emu_fault(const char *what, int on) {
    if (!strcmp(what, "no_device_guard")) {
        fault_no_device_guard = on != 0;
        return 0;
    }
    if (!strcmp(what, "drop_stream_waits")) {  // hipStreamWaitEvent becomes a no-op: what a program that forgot its waits does
        fault_drop_waits = on != 0;
        return 0;
    }
    if (!strcmp(what, "fail_malloc_at") || !strcmp(what, "fail_malloc_once")) {  // (0: disarm); restarts the call counter
        fault_malloc_tripped = false;
        fault_malloc_persistent = what[12] == 'a';
        fault_malloc_countdown = on > 0 ? on : 0;
        malloc_calls = 0;
        return 0;
    }
    if (!strcmp(what, "malloc_calls")) return (int)malloc_calls.load();  // hipMalloc + hipHostMalloc calls since the last arming
    if (!strcmp(what, "live_allocs")) {                                    // device + pinned allocations alive right now
        std::lock_guard<std::mutex> l(mu);
        return (int)allocs.size();
    }
    return 1;
}
int p2hot_emu_set_device(int d) { return hipSetDevice(d); }
int p2hot_emu_device_of(const void *p) { return device_of(p); }
}
