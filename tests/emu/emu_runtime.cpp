// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>
#include <vector>

namespace emu {

Fiber* cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;

namespace {
constexpr size_t kStackBytes = 96 * 1024;
constexpr size_t kLdsBytes = 160 * 1024;
alignas(64) char g_lds[kLdsBytes];

struct Wave {
  int alive = 0, arrived = 0, gen = 0;
  Xchg buf[2];
};

std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
std::vector<char*> g_stack_pool;
void* g_sched_sp = nullptr;
const std::function<void()>* g_body = nullptr;
int g_alive = 0, g_at_barrier = 0;

extern "C" void emu_switch(void** save_sp, void* new_sp);
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

void yield_to_scheduler() {
  Fiber* f = cur;
  emu_switch(&f->sp, g_sched_sp);
}

void release_wave(Wave& w, int wave_id) {
  w.arrived = 0;
  w.gen++;
  const int base = wave_id * 64;
  for (int l = 0; l < 64 && base + l < (int)g_fibers.size(); ++l)
    if (g_fibers[base + l].state == 2) g_fibers[base + l].state = 0;
}

void release_barrier() {
  g_at_barrier = 0;
  for (auto& f : g_fibers)
    if (f.state == 1) f.state = 0;
}

void fiber_main() {
  (*g_body)();
  Fiber* f = cur;
  f->state = 3;
  g_alive--;
  Wave& w = g_waves[f->wave];
  w.alive--;
  // lanes that left early no longer take part in collectives: real hardware would hang or read garbage
  // at a full-wave collective with exited lanes only when EXEC-masked lanes are *needed*; the kernels here
  // never rely on that, so treat the remaining lanes as the whole wave.
  if (w.alive > 0 && w.arrived == w.alive) release_wave(w, f->wave);
  if (g_alive > 0 && g_at_barrier == g_alive) release_barrier();
  emu_switch(&f->sp, g_sched_sp);
  fprintf(stderr, "emu: resumed a finished fiber\n");
  abort();
}

void prepare_stack(Fiber& f) {
  // layout (low -> high): r15 r14 r13 r12 rbx rbp | return address | pad ; base 16-byte aligned so that
  // after the six pops and the ret, rsp == 8 (mod 16) exactly as at a normal function entry
  uintptr_t top = reinterpret_cast<uintptr_t>(f.stack) + kStackBytes;
  top &= ~uintptr_t(15);
  uint64_t* sp = reinterpret_cast<uint64_t*>(top) - 8;   // 64 bytes below the top, 16-byte aligned
  for (int i = 0; i < 6; ++i) sp[i] = 0;
  sp[6] = reinterpret_cast<uint64_t>(&fiber_main);
  sp[7] = 0;
  f.sp = sp;
}
}  // namespace

void* dyn_lds() { return g_lds; }

void syncthreads() {
  Fiber* f = cur;
  g_at_barrier++;
  if (g_at_barrier == g_alive) {
    f->state = 0;
    release_barrier();
    return;
  }
  f->state = 1;
  yield_to_scheduler();
}

const Xchg& wave_exchange(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  Fiber* f = cur;
  Wave& w = g_waves[f->wave];
  const int par = w.gen & 1;
  uint32_t* slot = w.buf[par].w[f->lane];
  slot[0] = a; slot[1] = b; slot[2] = c; slot[3] = d;
  w.arrived++;
  if (w.arrived == w.alive) {
    release_wave(w, f->wave);
    f->state = 0;
  } else {
    f->state = 2;
    yield_to_scheduler();
  }
  return w.buf[par];
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
  if (cur != nullptr) { fprintf(stderr, "emu: nested launch\n"); abort(); }
  if (lds_bytes > kLdsBytes) { fprintf(stderr, "emu: %zu bytes of LDS requested (> 160 KiB)\n", lds_bytes); abort(); }
  const int threads = (int)(block.x * block.y * block.z);
  if (threads <= 0 || threads > 1024) { fprintf(stderr, "emu: bad block size %d\n", threads); abort(); }
  while ((int)g_stack_pool.size() < threads) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, kStackBytes) != 0) abort();
    g_stack_pool.push_back(static_cast<char*>(p));
  }
  g_blockDim = block;
  g_gridDim = grid;
  g_body = &body;
  const int n_waves = (threads + 63) / 64;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = dim3(bx, by, bz);
        // poison the dynamic LDS so reads of never-written LDS show up as NaNs
        memset(g_lds, 0xFF, lds_bytes ? lds_bytes : 64);
        g_fibers.assign(threads, Fiber());
        g_waves.assign(n_waves, Wave());
        for (int t = 0; t < threads; ++t) {
          Fiber& f = g_fibers[t];
          f.linear = t;
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          f.wave = t / 64;
          f.lane = t % 64;
          f.state = 0;
          f.stack = g_stack_pool[t];
          prepare_stack(f);
          g_waves[f.wave].alive++;
        }
        g_alive = threads;
        g_at_barrier = 0;
        while (g_alive > 0) {
          bool progressed = false;
          for (int t = 0; t < threads; ++t) {
            Fiber& f = g_fibers[t];
            if (f.state != 0) continue;
            progressed = true;
            cur = &f;
            emu_switch(&g_sched_sp, f.sp);
            cur = nullptr;
          }
          if (!progressed) {
            int nb = 0, nw = 0;
            for (auto& f : g_fibers) { nb += f.state == 1; nw += f.state == 2; }
            fprintf(stderr, "emu: DEADLOCK in block (%u,%u,%u): %d threads alive, %d at __syncthreads, %d at a wave "
                            "collective (divergent barrier / collective?)\n", bx, by, bz, g_alive, nb, nw);
            abort();
          }
        }
      }
  g_body = nullptr;
}

}  // namespace emu
