// tests/emu/hipemu.cpp -- TEST INFRASTRUCTURE ONLY: runtime of the SIMT emulator (see hipemu.h).
#include "hipemu.h"
#include <stdlib.h>
#include <stdio.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
#include <atomic>
#include <mutex>

extern "C" unsigned char __start_hipemu_lds[], __stop_hipemu_lds[];      // the section of all `__shared__` variables of the library (hipemu.h)
static unsigned char hipemu_lds_anchor __attribute__((section("hipemu_lds"), used)) = 0;      // (the section exists even in a library without kernels)

namespace hipemu {

thread_local Fiber* cur = nullptr;
thread_local BlockCtx* blk = nullptr;
static thread_local void* sched_sp = nullptr;
static thread_local kernel_thunk_t g_thunk = nullptr;
static thread_local void* g_args = nullptr;

// void hipemu_switch(void** save_sp, void* load_sp): save callee-saved regs + sp, load the other side.
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
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
.size hipemu_switch,.-hipemu_switch
)");

static const size_t STACK_BYTES = 256 * 1024;

static void fiber_entry()
{
    g_thunk(g_args);
    cur->state = 3;
    blk->live--;
    blk->waves[cur->tid >> 6].live--;
    // a finished lane may complete a rendezvous others are waiting on
    if (blk->live && blk->arrived == blk->live) { blk->arrived = 0; blk->gen++; }
    WaveCtx& w = blk->waves[cur->tid >> 6];
    if (w.live && w.arrived == w.live) { w.arrived = 0; w.gen++; }
    hipemu_switch(&cur->sp, sched_sp);
    abort();  // never resumed
}

void yield_to_scheduler() { hipemu_switch(&cur->sp, sched_sp); }

void block_barrier()
{
    BlockCtx* b = blk;
    unsigned long g = b->gen;
    // the last arrival releases the others but yields too, so that everybody resumes in thread order (lane 0 first), the
    // order in which the hardware serves the lanes of one LDS instruction
    if (++b->arrived == b->live) { b->arrived = 0; b->gen++; yield_to_scheduler(); return; }
    cur->state = 1; cur->wait_gen = g;
    yield_to_scheduler();
}

void wave_barrier()
{
    WaveCtx& w = wv();
    unsigned long g = w.gen;
    if (++w.arrived == w.live) { w.arrived = 0; w.gen++; yield_to_scheduler(); return; }
    cur->state = 2; cur->wait_gen = g;
    yield_to_scheduler();
}

struct Pool {   // per OS thread: fiber stacks reused across workgroups
    char* mem = nullptr; size_t n = 0;
    char* get(size_t nthreads)
    {
        if (nthreads > n) {
            if (mem) munmap(mem, n * STACK_BYTES);
            mem = (char*)mmap(nullptr, nthreads * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (mem == MAP_FAILED) { perror("hipemu mmap"); abort(); }
            n = nthreads;
        }
        return mem;
    }
    ~Pool() { if (mem) munmap(mem, n * STACK_BYTES); }
};

static void run_block(dim3 grid, dim3 block, unsigned linear_block, kernel_thunk_t thunk, void* args, Pool& pool)
{
    BlockCtx b;
    unsigned nthreads = block.x * block.y * block.z;
    b.bdim = { block.x, block.y, block.z }; b.gdim = { grid.x, grid.y, grid.z };
    b.bidx = { linear_block % grid.x, (linear_block / grid.x) % grid.y, linear_block / (grid.x * grid.y) };
    b.nthreads = nthreads; b.nwaves = (nthreads + 63) / 64;
    b.gen = 0; b.arrived = 0; b.live = nthreads;
    std::vector<Fiber> fibers(nthreads);
    std::vector<WaveCtx> waves(b.nwaves);
    b.fibers = fibers.data(); b.waves = waves.data();
    char* stacks = pool.get(nthreads);
    for (unsigned w = 0; w < b.nwaves; w++) { waves[w].gen = 0; waves[w].arrived = 0; waves[w].live = std::min(64u, nthreads - w * 64); }
    for (unsigned t = 0; t < nthreads; t++) {
        Fiber& f = fibers[t];
        f.tid = t; f.tidx = { t % block.x, (t / block.x) % block.y, t / (block.x * block.y) };
        f.state = 0; f.stack = stacks + (size_t)t * STACK_BYTES;
        // initial frame: 6 callee-saved slots + return address (fiber_entry); keep 16-byte ABI alignment at entry
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8);       // so that after `ret` rsp % 16 == 8 as at a normal call entry
        *--sp = (void*)&fiber_entry;
        for (int i = 0; i < 6; i++) *--sp = nullptr;
        f.sp = sp;
    }
    blk = &b; g_thunk = thunk; g_args = args;
    unsigned remaining = nthreads;
    while (remaining) {
        bool progressed = false;
        for (unsigned t = 0; t < nthreads; t++) {
            Fiber& f = fibers[t];
            if (f.state == 3) continue;
            if (f.state == 1) { if (b.gen == f.wait_gen) continue; f.state = 0; }
            else if (f.state == 2) { if (waves[t >> 6].gen == f.wait_gen) continue; f.state = 0; }
            cur = &f;
            hipemu_switch(&sched_sp, f.sp);
            progressed = true;
            if (f.state == 3) remaining--;
        }
        if (!progressed) { fprintf(stderr, "hipemu: deadlock (divergent barrier / collective) in block %u\n", linear_block); abort(); }
    }
    blk = nullptr; cur = nullptr;
}

void launch(dim3 grid, dim3 block, kernel_thunk_t thunk, void* args, int)
{
    // `__shared__` variables are plain statics, so workgroups of one process run one after another;
    // tests that want parallelism split the input across processes (frames are independent).
    // One launch at a time per process for the same reason: host threads that own different contexts (gc_multi.hip) take turns.
    static std::mutex launchMutex;
    std::lock_guard<std::mutex> guard(launchMutex);
    unsigned nblocks = grid.x * grid.y * grid.z;
    static Pool pool;
    static const char* poison = getenv("GC_EMU_POISON_LDS");
    static unsigned long long px = poison ? 0x9E3779B97F4A7C15ull * (unsigned long long)(atoi(poison) + 1) : 0ull;
    for (unsigned i = 0; i < nblocks; i++) {
        if (poison) for (unsigned char* q = __start_hipemu_lds; q < __stop_hipemu_lds; q++) { px ^= px << 13; px ^= px >> 7; px ^= px << 17; *q = (unsigned char)(px >> 32); }
        run_block(grid, block, i, thunk, args, pool);
    }
}

}  // namespace hipemu
