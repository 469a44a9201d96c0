// Fiber scheduler of the host SIMT emulator (see tools/emu/hip/hip_runtime.h).
// DEVELOPMENT/TEST TOOLING ONLY -- never linked into the product library.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

namespace emu {
Block* g_blk = nullptr;
long g_mfma_count = 0;

// Minimal x86-64 SysV context switch: saves callee-saved registers on the current stack,
// stores rsp to *save_sp, switches to load_sp and restores.
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

static void fiber_entry() {
    Block* b = g_blk;
    b->body();
    b->lanes[b->cur].state = 3;
    emu_switch(&b->lanes[b->cur].sp, b->sched_sp);
    abort();  // a finished fiber is never resumed
}

void yield_to_scheduler(int state) {
    Block* b = g_blk;
    Lane& l = b->lanes[b->cur];
    l.state = state;
    emu_switch(&l.sp, b->sched_sp);
}

static constexpr size_t STACK_BYTES = 1u << 20;

static void run_block(Block& b) {
    const int nthreads = (int)(b.bdim.x * b.bdim.y * b.bdim.z);
    b.lanes.assign((size_t)nthreads, Lane());
    static std::vector<char*> stacks;
    while ((int)stacks.size() < nthreads) {
        void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("mmap"); abort(); }
        stacks.push_back(static_cast<char*>(p));
    }
    for (int i = 0; i < nthreads; ++i) {
        Lane& l = b.lanes[(size_t)i];
        l.tid = dim3(i % b.bdim.x, (i / b.bdim.x) % b.bdim.y, i / (b.bdim.x * b.bdim.y));
        l.stack = stacks[(size_t)i];
        // initial frame: six callee-saved slots + return address (fiber_entry); keep the ABI's
        // "rsp % 16 == 8 at function entry" by placing the return address at a 16-byte boundary - 8.
        uintptr_t top = (reinterpret_cast<uintptr_t>(l.stack) + STACK_BYTES) & ~uintptr_t(15);
        void** sp = reinterpret_cast<void**>(top - 8);
        *sp = nullptr;                                       // fake return address of fiber_entry's "caller"
        *--sp = reinterpret_cast<void*>(&fiber_entry);       // popped by 'ret' -> rsp = top-8 (== 8 mod 16)
        for (int k = 0; k < 6; ++k) *--sp = nullptr;
        l.sp = sp;
        l.state = 0;
    }
    const int nwaves = (nthreads + 63) / 64;
    g_blk = &b;
    for (;;) {
        int done_waves = 0, at_block = 0;
        for (int w = 0; w < nwaves; ++w) {
            const int lo = w * 64, hi = std::min(nthreads, lo + 64);
            // run this wave until every lane sits at a block barrier or has finished
            for (;;) {
                int n_wave = 0, n_block = 0, n_done = 0, n_spin = 0;
                for (int i = lo; i < hi; ++i) {
                    Lane& l = b.lanes[(size_t)i];
                    if (l.state == 0 || l.state == 1) {
                        l.state = 0;
                        b.cur = i;
                        emu_switch(&b.sched_sp, l.sp);
                    }
                    n_wave += l.state == 1;
                    n_block += l.state == 2;
                    n_done += l.state == 3;
                    n_spin += l.state == 4;
                }
                const int n = hi - lo;
                if (n_wave == n) continue;                      // all at the same wave collective: release
                if (n_spin == n) {                              // the whole wave polls a flag: give the other waves a turn
                    for (int i = lo; i < hi; ++i) b.lanes[(size_t)i].state = 0;
                    break;
                }
                if (n_done == n) { ++done_waves; break; }
                if (n_block == n) { ++at_block; break; }
                if (n_wave + n_done == n && n_wave > 0) {
                    fprintf(stderr, "emu: wave %d diverged at a wave collective (%d waiting, %d finished)\n", w, n_wave, n_done);
                    abort();
                }
                if (n_block + n_done == n) { ++at_block; break; }   // some lanes exited early: barrier counts the rest
                fprintf(stderr, "emu: wave %d has lanes at different sync kinds (wave %d block %d done %d)\n", w, n_wave, n_block, n_done);
                abort();
            }
        }
        if (done_waves == nwaves) break;
        if (done_waves + at_block == nwaves) {
            for (Lane& l : b.lanes) if (l.state == 2) l.state = 0;   // release the block barrier
            continue;
        }
    }
    g_blk = nullptr;
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
    Block b;
    b.bdim = block;
    b.gdim = grid;
    b.body = body;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                b.bid = dim3(x, y, z);
                run_block(b);
            }
}
}  // namespace emu
