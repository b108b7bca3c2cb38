// hipemu scheduler: one OS thread, one ucontext fiber per GPU thread, workgroups run one after another.
// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h).
#include "hip/hip_runtime.h"

#include <sys/mman.h>
#include <ucontext.h>
#include <vector>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;
char* hipemu_dynamic_lds = nullptr;

namespace {
enum State { RUN, AT_BARRIER, AT_WAVE, DONE };
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    State st = RUN;
    uint3_emu tid;
    int lane, wave;
    unsigned wave_ops = 0;
};
constexpr size_t STACK = 256 * 1024;
std::vector<Fiber> fibers;
ucontext_t sched_ctx;
Fiber* cur = nullptr;
const std::function<void()>* cur_body = nullptr;
// [wave][parity][lane][16 words]
std::vector<uint32_t> exch;
std::vector<char*> stack_pool;
char* lds_mem = nullptr;
constexpr size_t LDS_MAX = 160 * 1024;

void yield(State s) {
    cur->st = s;
    swapcontext(&cur->ctx, &sched_ctx);
}
void trampoline() {
    (*cur_body)();
    cur->st = DONE;
    swapcontext(&cur->ctx, &sched_ctx);
}
[[noreturn]] void die(const char* msg) {
    fprintf(stderr, "hipemu: %s (block %u,%u thread %u)\n", msg, blockIdx.x, blockIdx.y, cur ? cur->tid.x : 0u);
    abort();
}
}  // namespace

int hipemu_lane() { return cur->lane; }

void hipemu_syncthreads() { yield(AT_BARRIER); }

const uint32_t* hipemu_wave_exchange(const uint32_t* words, int n) {
    unsigned par = cur->wave_ops & 1u;
    uint32_t* tab = &exch[((size_t)cur->wave * 2 + par) * 64 * 16];
    for (int i = 0; i < n; ++i) tab[cur->lane * 16 + i] = words[i];
    cur->wave_ops++;
    yield(AT_WAVE);
    return tab;
}

void hipemu_launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes) {
    size_t nthreads = (size_t)block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > 1024) die("bad block size");
    if (lds_bytes > LDS_MAX) die("dynamic LDS request exceeds 160 KiB");
    if (!lds_mem) {
        // LDS followed by a PROT_NONE guard page so that overruns of the dynamic region fault
        lds_mem = (char*)mmap(nullptr, LDS_MAX + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        mprotect(lds_mem + LDS_MAX, 4096, PROT_NONE);
    }
    size_t nwaves = (nthreads + 63) / 64;
    while (stack_pool.size() < nthreads) stack_pool.push_back((char*)malloc(STACK));
    exch.assign(nwaves * 2 * 64 * 16, 0);
    blockDim = block;
    gridDim = grid;
    cur_body = &body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        // place the dynamic region so that its END abuts the guard page (catches overruns of the request)
        size_t rounded = (lds_bytes + 15) & ~(size_t)15;
        hipemu_dynamic_lds = lds_mem + LDS_MAX - rounded;
        memset(lds_mem, 0xCD, LDS_MAX);   // poison: uninitialised LDS reads give large garbage, not zeros
        fibers.assign(nthreads, Fiber());
        for (size_t t = 0; t < nthreads; ++t) {
            Fiber& f = fibers[t];
            f.tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y))};
            f.lane = (int)(t & 63);
            f.wave = (int)(t >> 6);
            f.stack = stack_pool[t];
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, trampoline, 0);
        }
        size_t done = 0;
        while (done < nthreads) {
            bool progressed = false;
            for (size_t t = 0; t < nthreads; ++t) {
                Fiber& f = fibers[t];
                if (f.st != RUN) continue;
                cur = &f;
                threadIdx = f.tid;
                swapcontext(&sched_ctx, &f.ctx);
                progressed = true;
                if (f.st == DONE) done++;
            }
            // release waves whose (non-exited) lanes all arrived at the same wave op
            for (size_t w = 0; w < nwaves; ++w) {
                size_t lo = w * 64, hi = std::min(nthreads, lo + 64);
                bool all = true, any = false;
                unsigned ops = 0;
                for (size_t t = lo; t < hi; ++t) {
                    if (fibers[t].st == AT_WAVE) { if (!any) ops = fibers[t].wave_ops; any = true; if (fibers[t].wave_ops != ops) all = false; }
                    else if (fibers[t].st != DONE) all = false;   // exited lanes do not take part (as on hardware)
                }
                if (any && all) { for (size_t t = lo; t < hi; ++t) fibers[t].st = RUN; progressed = true; }
            }
            // release the workgroup barrier when every live thread is at it
            size_t at_bar = 0;
            for (auto& f : fibers) if (f.st == AT_BARRIER) at_bar++;
            if (at_bar && at_bar + done == nthreads) { for (auto& f : fibers) if (f.st == AT_BARRIER) f.st = RUN; progressed = true; }
            if (!progressed && done < nthreads) {
                cur = &fibers[0];
                die("deadlock: divergent __syncthreads / wave op (a cross-lane op needs all 64 lanes, a barrier all threads)");
            }
        }
    }
    cur = nullptr;
}
