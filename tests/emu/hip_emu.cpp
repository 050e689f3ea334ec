// TEST INFRASTRUCTURE — fiber scheduler for tests/emu/hip_emu.h (see the header for scope).
#include "hip_emu.h"

#include <sys/mman.h>
#include <unistd.h>

#ifdef MF_EMU_FAST_SWITCH
// void mf_emu_ctx_switch(Ctx* from, Ctx* to): push the callee-saved registers, park the stack pointer in *from, adopt
// *to's, pop its callee-saved registers and return into it.
asm(R"(
    .text
    .globl mf_emu_ctx_switch
    .type mf_emu_ctx_switch, @function
mf_emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size mf_emu_ctx_switch, .-mf_emu_ctx_switch
)");
#endif

namespace emu {

BlockState* g_block = nullptr;
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

static void fiber_entry() {
    BlockState* b = g_block;
    MF_FIBER_FINISH(nullptr, &b->sched_bottom, &b->sched_size);      // first entry: learn the scheduler's stack
    b->body();
    b->fibers[b->cur].done = true;
    MF_FIBER_START(nullptr, b->sched_bottom, b->sched_size);         // this fiber never runs again
#ifdef MF_EMU_FAST_SWITCH
    mf_emu_ctx_switch(&b->fibers[b->cur].ctx, &b->sched);
#else
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
#endif
    abort();                                                         // not reached
}

// Guard-paged dynamic LDS: [ PROT_NONE page | slack ... block (size bytes, 64-byte aligned start) | PROT_NONE page ]
struct GuardedLds {
    char* map = nullptr;
    size_t map_bytes = 0;
    char* block = nullptr;
    GuardedLds(size_t size) {
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        const size_t body = ((size + 63) / 64 * 64 + page - 1) / page * page;
        map_bytes = body + 2 * page;
        map = (char*)mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (map == MAP_FAILED) { perror("emu: mmap"); abort(); }
        mprotect(map, page, PROT_NONE);
        mprotect(map + page + body, page, PROT_NONE);
        // the kernels' MF_DYN_SMEM rounds the pointer UP to 64 bytes: choose a start that is already aligned and whose
        // end is within 63 bytes of the upper guard page (sizes are multiples of 4, images mostly of 64)
        block = map + page + body - (size + 63) / 64 * 64;
#ifdef MF_EMU_ASAN
        __asan_poison_memory_region(map + page, (size_t)(block - (map + page)));
        __asan_poison_memory_region(block + size, (size_t)((map + page + body) - (block + size)));
#endif
    }
    ~GuardedLds() {
#ifdef MF_EMU_ASAN
        __asan_unpoison_memory_region(map, map_bytes);
#endif
        munmap(map, map_bytes);
    }
};

void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    BlockState bs;
    bs.body = std::move(body);
    bs.fibers.resize(nthreads);
    bs.waves.resize((nthreads + WAVE - 1) / WAVE);
    // fiber stacks come from a pool that lives as long as the process: a launch of 1024 threads would otherwise map and
    // unmap 512 MiB every time (minutes of kernel time over a test run, more under ASan)
    static std::vector<char*> stack_pool;
    while ((int)stack_pool.size() < nthreads) stack_pool.push_back((char*)malloc(STACK));
    for (int i = 0; i < nthreads; ++i) bs.fibers[i].stack = stack_pool[i];
    GuardedLds dyn(smem);
    bs.dyn_smem = dyn.block;
    BlockState* prev = g_block;
    g_block = &bs;
    g_blockDim = block;
    g_gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = dim3(bx, by, bz);
                bs.barrier_arrived = 0;
                for (auto& w : bs.waves) w.arrived = 0;
                int t = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
                            Fiber& f = bs.fibers[t];
                            f.done = false;
                            f.tid = dim3(tx, ty, tz);
#ifdef MF_EMU_FAST_SWITCH
                            {   // initial frame: six zeroed callee-saved registers, then fiber_entry as the return address,
                                // then a null return address for fiber_entry itself (it never returns); the stack pointer
                                // is congruent to 8 mod 16 when fiber_entry starts, as after a call
                                uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
                                void** sp = (void**)top;
                                *--sp = nullptr;
                                *--sp = (void*)fiber_entry;
                                for (int r = 0; r < 6; ++r) *--sp = nullptr;
                                f.ctx.sp = sp;
                            }
#else
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = f.stack;
                            f.ctx.uc_stack.ss_size = STACK;
                            f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
                        }
                int remaining = nthreads;
                long spins = 0;
                while (remaining > 0) {
                    int progressed = 0;
                    for (int i = 0; i < nthreads; ++i) {
                        Fiber& f = bs.fibers[i];
                        if (f.done) continue;
                        bs.cur = i;
                        g_threadIdx = f.tid;
                        MF_FIBER_START(&bs.sched_fake_stack, f.stack, STACK);
#ifdef MF_EMU_FAST_SWITCH
                        mf_emu_ctx_switch(&bs.sched, &f.ctx);
#else
                        swapcontext(&bs.sched, &f.ctx);
#endif
                        MF_FIBER_FINISH(bs.sched_fake_stack, nullptr, nullptr);
                        if (f.done) { --remaining; ++progressed; }
                    }
                    if (++spins > 200000000L) { fprintf(stderr, "emu: deadlock (divergent collective?)\n"); abort(); }
                }
            }
    g_block = prev;
}

}  // namespace emu
