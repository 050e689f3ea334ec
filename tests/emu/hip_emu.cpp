// TEST INFRASTRUCTURE — fiber scheduler for tests/emu/hip_emu.h (see the header for scope).
#include "hip_emu.h"

namespace emu {

BlockState* g_block = nullptr;
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

static void fiber_entry() {
    BlockState* b = g_block;
    b->body();
    b->fibers[b->cur].done = true;
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    BlockState bs;
    bs.body = std::move(body);
    bs.fibers.resize(nthreads);
    bs.waves.resize((nthreads + WAVE - 1) / WAVE);
    for (auto& f : bs.fibers) f.stack = (char*)malloc(STACK);
    std::vector<char> dyn(smem + 64);
    bs.dyn_smem = dyn.data();
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
                            getcontext(&f.ctx);
                            f.ctx.uc_stack.ss_sp = f.stack;
                            f.ctx.uc_stack.ss_size = STACK;
                            f.ctx.uc_link = nullptr;
                            makecontext(&f.ctx, (void (*)())fiber_entry, 0);
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
                        swapcontext(&bs.sched, &f.ctx);
                        if (f.done) { --remaining; ++progressed; }
                    }
                    if (++spins > 200000000L) { fprintf(stderr, "emu: deadlock (divergent collective?)\n"); abort(); }
                }
            }
    for (auto& f : bs.fibers) free(f.stack);
    g_block = prev;
}

}  // namespace emu
