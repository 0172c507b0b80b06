// hipemu runtime: fiber scheduler + host API stubs.  DEVELOPMENT / TEST TOOL ONLY
// (see tools/hipemu/include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <mutex>
#include <thread>

extern "C" void hipemu_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");

struct hipemu_stream {
    bool capturing = false;
    std::vector<std::function<void()>> recorded;
};
struct hipemu_graph {
    std::vector<std::function<void()>> ops;
};

namespace hipemu {

thread_local BlockCtx* g_blk = nullptr;
thread_local Fiber* g_cur = nullptr;

static const size_t kStackSize = 256 * 1024;
static hipemu_stream g_default_stream;
static int g_num_workers = 0;

static int num_workers() {
    if (g_num_workers == 0) {
        const char* e = getenv("HIPEMU_THREADS");
        int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        g_num_workers = n < 1 ? 1 : (n > 64 ? 64 : n);
    }
    return g_num_workers;
}

void yield_to_scheduler(int new_state) {
    Fiber* f = g_cur;
    f->state = new_state;
    hipemu_switch(&f->sp, g_blk->sched_sp);
}

static void fiber_entry() {
    (*g_blk->body)();
    Fiber* f = g_cur;
    f->state = F_DONE;
    for (;;) hipemu_switch(&f->sp, g_blk->sched_sp);
}

static void init_fiber(Fiber& f, char* stack) {
    f.stack = stack;
    uintptr_t top = ((uintptr_t)stack + kStackSize) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // fake return address of fiber_entry (keeps rsp%16==8 at entry)
    *--sp = (void*)&fiber_entry;     // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12..r15
    f.sp = (void*)sp;
    f.state = F_READY;
}

static void run_fiber(BlockCtx& b, Fiber& f) {
    g_cur = &f;
    hipemu_switch(&b.sched_sp, f.sp);
    g_cur = nullptr;
}

static void run_block(BlockCtx& b, unsigned nthreads) {
    const unsigned nwaves = (nthreads + 63) / 64;
    for (unsigned t = 0; t < nthreads; ++t) init_fiber(b.fibers[t], b.stack_pool + (size_t)t * kStackSize);
    for (unsigned w = 0; w < nwaves; ++w) memset(b.waves[w].part, 0, sizeof(b.waves[w].part));
    for (;;) {
        bool all_done = true;
        for (unsigned w = 0; w < nwaves; ++w) {
            const unsigned lo = w * 64, hi = (lo + 64 < nthreads) ? lo + 64 : nthreads;
            for (;;) {
                bool ran = false;
                for (unsigned t = lo; t < hi; ++t)
                    if (b.fibers[t].state == F_READY) { run_fiber(b, b.fibers[t]); ran = true; }
                // wave-level rendezvous: every live lane is parked and at least one waits for the wave
                bool any_wave = false, any_ready = false;
                for (unsigned t = lo; t < hi; ++t) {
                    if (b.fibers[t].state == F_WAIT_WAVE) any_wave = true;
                    if (b.fibers[t].state == F_READY) any_ready = true;
                }
                if (any_ready) continue;
                if (any_wave) {
                    for (unsigned t = lo; t < hi; ++t)
                        if (b.fibers[t].state == F_WAIT_WAVE) b.fibers[t].state = F_READY;
                    continue;
                }
                (void)ran;
                break;
            }
        }
        bool any_block = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            if (b.fibers[t].state != F_DONE) all_done = false;
            if (b.fibers[t].state == F_WAIT_BLOCK) any_block = true;
        }
        if (all_done) break;
        if (!any_block) { fprintf(stderr, "hipemu: scheduler deadlock\n"); abort(); }
        for (unsigned t = 0; t < nthreads; ++t)
            if (b.fibers[t].state == F_WAIT_BLOCK) b.fibers[t].state = F_READY;
    }
}

static void execute(dim3 grid, dim3 block, const std::function<void()>& body) {
    const unsigned nthreads = block.x * block.y * block.z;
    const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
    if (nthreads == 0 || nblocks == 0) return;
    if (nthreads > 1024) { fprintf(stderr, "hipemu: block too large (%u)\n", nthreads); abort(); }
    std::atomic<unsigned long long> next{0};
    auto worker = [&]() {
        BlockCtx b;
        b.fibers.resize(nthreads);
        b.waves.resize((nthreads + 63) / 64);
        b.stack_pool_sz = (size_t)nthreads * kStackSize;
        b.stack_pool = (char*)mmap(nullptr, b.stack_pool_sz, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (b.stack_pool == (char*)MAP_FAILED) { perror("hipemu mmap"); abort(); }
        b.body = &body;
        b.bdim = {block.x, block.y, block.z};
        b.gdim = {grid.x, grid.y, grid.z};
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = b.fibers[t];
            f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
            f.lane = (int)(t & 63);
            f.wave = (int)(t >> 6);
        }
        g_blk = &b;
        for (;;) {
            unsigned long long i = next.fetch_add(1);
            if (i >= nblocks) break;
            b.bid = {(unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((unsigned long long)grid.x * grid.y))};
            run_block(b, nthreads);
        }
        g_blk = nullptr;
        munmap(b.stack_pool, b.stack_pool_sz);
    };
    unsigned nw = (unsigned)num_workers();
    if (nblocks < nw) nw = (unsigned)nblocks;
    if (nw <= 1) { std::thread th(worker); th.join(); return; }   // own thread: keeps TLS/stack usage off the caller
    std::vector<std::thread> ths;
    for (unsigned i = 0; i < nw; ++i) ths.emplace_back(worker);
    for (auto& t : ths) t.join();
}

void enqueue(hipStream_t stream, std::function<void()> op) {
    hipemu_stream* s = stream ? stream : &g_default_stream;
    if (s->capturing) s->recorded.push_back(std::move(op));
    else op();
}

void launch(dim3 grid, dim3 block, hipStream_t stream, std::function<void()> body) {
    enqueue(stream, [grid, block, body]() { execute(grid, block, body); });
}

}  // namespace hipemu

hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess (emu)" : "hipError (emu)"; }
hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t s) {
    hipemu::enqueue(s, [=]() { memset(p, value, bytes); });
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    hipemu::enqueue(s, [=]() { memmove(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
    hipemu_stream* st = s ? s : &hipemu::g_default_stream;
    if (st->capturing) return hipErrorInvalidValue;
    st->capturing = true;
    st->recorded.clear();
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* graph) {
    hipemu_stream* st = s ? s : &hipemu::g_default_stream;
    if (!st->capturing) return hipErrorInvalidValue;
    st->capturing = false;
    auto* g = new hipemu_graph();
    g->ops.swap(st->recorded);
    *graph = g;
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, void*, void*, size_t) {
    auto* g = new hipemu_graph();
    g->ops = graph->ops;
    *exec = g;
    return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t s) {
    for (auto& op : exec->ops) hipemu::enqueue(s, op);
    return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t exec) { delete exec; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t graph) { delete graph; return hipSuccess; }
