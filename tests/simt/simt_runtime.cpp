// TEST INFRASTRUCTURE ONLY (see include/hip/hip_runtime.h).  Fiber scheduler of the lane-by-lane kernel emulation.
//
// One OS thread.  Workgroups run one after the other; inside a workgroup every HIP thread is a fiber with its own stack.  A fiber runs
// until it reaches a meeting point (wave_sync / block_sync); the last lane to arrive releases the others and keeps running.
// A launch whose fibers all wait and none can be released is reported as a deadlock (divergent cross-lane operation or barrier).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <vector>
#include <algorithm>

extern "C" void simt_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
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
.size simt_switch,.-simt_switch
)");

namespace simt {

enum State { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Wave {
  int live, arrived;
  void (*then_fn)(void*);
  void* then_ctx;
  uint64_t x[2][64][8];                     // 64 bytes per lane: two 8 x fp32 MFMA operands in the fp32-operand build
  float tile[2][256];
  float tile32[1024];
};

struct Fiber {
  void* sp;
  char* stack;
  uint3_ tid;
  int lin, lane, wave, state;
  long seq;
};

Fiber* cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;
int g_error = 0;

static const size_t STACK = 512 << 10;
static std::vector<Fiber> fibers;
static std::vector<char*> stacks;
static std::vector<Wave> waves;
static std::vector<int> runq;
static size_t rq_head = 0;
static void* sched_sp;
static int block_live, block_arrived;
static std::vector<char> smem;
static BodyFn body;
static void* body_ctx;
// Order in which runnable fibers are resumed: 0 = ascending thread id, 1 = descending, 2 = pseudo-random (seeded).  A kernel
// without data races gives the same result under every order; a missing barrier between a producer and a consumer wave shows up
// as a difference (the fixed ascending order alone would always run the producer first).
static int sched_mode = 0;
static uint64_t sched_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_rand() {
  sched_state ^= sched_state << 13; sched_state ^= sched_state >> 7; sched_state ^= sched_state << 17;
  return sched_state;
}
static void push_batch(std::vector<int>& ids) {           // ids: fibers that became runnable together
  if (sched_mode == 1) std::reverse(ids.begin(), ids.end());
  else if (sched_mode == 2)
    for (size_t i = ids.size(); i > 1; --i) std::swap(ids[i - 1], ids[next_rand() % i]);
  runq.insert(runq.end(), ids.begin(), ids.end());
}

const uint3_& tid() { return cur->tid; }
int lane() { return cur->lane; }
void* dyn_smem() { return (void*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63); }
uint64_t* xslot(int l, int buf) { return waves[cur->wave].x[buf][l]; }
int next_buf() { return (cur->seq++) & 1; }
long op_seq() { return (long)cur->seq; }

static void yield_to_scheduler() { simt_switch(&cur->sp, sched_sp); }

static void release_wave(int w) {
  Wave& wv = waves[w];
  wv.arrived = 0;
  std::vector<int> ids;
  for (int l = 0; l < 64; ++l) {
    size_t i = (size_t)w * 64 + l;
    if (i < fibers.size() && fibers[i].state == WAIT_WAVE) {
      fibers[i].state = RUNNABLE;
      ids.push_back((int)i);
    }
  }
  push_batch(ids);
}
static void release_block() {
  block_arrived = 0;
  std::vector<int> ids;
  for (auto& f : fibers)
    if (f.state == WAIT_BLOCK) {
      f.state = RUNNABLE;
      ids.push_back(f.lin);
    }
  push_batch(ids);
}

float* wave_tile(int buf) { return waves[cur->wave].tile[buf]; }
float* wave_tile32() { return waves[cur->wave].tile32; }

static void complete_wave(int w) {                       // every live lane has arrived
  Wave& wv = waves[w];
  if (wv.then_fn) {
    void (*fn)(void*) = wv.then_fn;
    wv.then_fn = nullptr;
    fn(wv.then_ctx);
  }
  release_wave(w);
}

void wave_sync_then(void (*fn)(void*), void* ctx) {
  Wave& wv = waves[cur->wave];
  wv.then_fn = fn;
  wv.then_ctx = ctx;
  if (++wv.arrived == wv.live) {
    complete_wave(cur->wave);
    return;
  }
  cur->state = WAIT_WAVE;
  yield_to_scheduler();
}

void wave_sync() {
  Wave& wv = waves[cur->wave];
  if (++wv.arrived == wv.live) {
    complete_wave(cur->wave);
    return;
  }
  cur->state = WAIT_WAVE;
  yield_to_scheduler();
}

void block_sync() {
  if (++block_arrived == block_live) {
    release_block();
    return;
  }
  cur->state = WAIT_BLOCK;
  yield_to_scheduler();
}

static void fiber_main() {
  body(body_ctx);
  Fiber* f = cur;
  f->state = DONE;
  Wave& wv = waves[f->wave];
  --wv.live;
  --block_live;
  if (wv.live > 0 && wv.arrived == wv.live) {
    Fiber* self = cur;                                  // the completion callback addresses the wave through `cur`
    complete_wave(f->wave);
    cur = self;
  }
  if (block_live > 0 && block_arrived == block_live) release_block();
  yield_to_scheduler();
  abort();                                           // a finished fiber is never resumed
}

static char* get_stack(size_t i) {
  while (stacks.size() <= i) {
    void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("simt: mmap"); abort(); }
    stacks.push_back((char*)p);
  }
  return stacks[i];
}

static size_t smem_bytes = 0;                       // dynamic LDS of the current launch
static size_t lds_limit = 160u << 10;               // 160 KB per workgroup on gfx950; the fp32-operand build doubles every 16-bit tile

static bool run_block(unsigned nthreads) {
  // A workgroup starts with UNDEFINED LDS contents on the device.  Here every byte of the dynamic LDS is set to 0xFF first (a NaN as
  // fp16 / bf16 / fp32, -1 as an integer): a kernel whose result depends on LDS it never wrote fails its parity check instead of
  // passing on the leftovers of the previous workgroup.
  if (smem_bytes) memset(dyn_smem(), 0xFF, smem_bytes);
  fibers.assign(nthreads, Fiber());
  const unsigned nw = (nthreads + 63) / 64;
  waves.assign(nw, Wave());
  runq.clear();
  rq_head = 0;
  block_live = (int)nthreads;
  block_arrived = 0;
  for (unsigned i = 0; i < nthreads; ++i) {
    Fiber& f = fibers[i];
    f.lin = (int)i;
    f.tid.x = i % g_blockDim.x;
    f.tid.y = (i / g_blockDim.x) % g_blockDim.y;
    f.tid.z = i / (g_blockDim.x * g_blockDim.y);
    f.lane = (int)(i & 63);
    f.wave = (int)(i >> 6);
    f.state = RUNNABLE;
    f.seq = 0;
    f.stack = get_stack(i);
    uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
    void** s = (void**)(top - 64);
    for (int k = 0; k < 6; ++k) s[k] = nullptr;
    s[6] = (void*)&fiber_main;
    s[7] = nullptr;
    f.sp = (void*)s;
    waves[f.wave].live++;
  }
  {
    std::vector<int> ids(nthreads);
    for (unsigned i = 0; i < nthreads; ++i) ids[i] = (int)i;
    push_batch(ids);
  }
  while (true) {
    if (rq_head == runq.size()) break;
    Fiber* f = &fibers[runq[rq_head++]];
    if (rq_head > 4096 && rq_head * 2 > runq.size()) {            // compact the queue
      runq.erase(runq.begin(), runq.begin() + (long)rq_head);
      rq_head = 0;
    }
    if (f->state != RUNNABLE) continue;
    cur = f;
    simt_switch(&sched_sp, f->sp);
  }
  cur = nullptr;
  if (block_live != 0) {
    int ww = 0, wb = 0;
    for (auto& f : fibers) { ww += f.state == WAIT_WAVE; wb += f.state == WAIT_BLOCK; }
    fprintf(stderr, "simt: DEADLOCK in block (%u,%u,%u): %d threads alive, %d at a wave meeting point, %d at __syncthreads\n",
            g_blockIdx.x, g_blockIdx.y, g_blockIdx.z, block_live, ww, wb);
    return false;
  }
  return true;
}

void launch(dim3 grid, dim3 block, size_t shmem, BodyFn fn, void* ctx) {
  g_gridDim = grid;
  g_blockDim = block;
  body = fn;
  body_ctx = ctx;
  if (smem.size() < shmem + 64) smem.resize(shmem + 64);
  smem_bytes = shmem;
  const unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > 1024 || shmem > lds_limit) { g_error = 1; return; }       // hipErrorInvalidValue
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        g_blockIdx = dim3(x, y, z);
        if (!run_block(nthreads)) { g_error = hipErrorLaunchFailure; return; }
      }
}

}  // namespace simt

// test control: order in which runnable fibers are resumed (0 ascending, 1 descending, 2 pseudo-random with `seed`)
extern "C" void simt_set_lds_limit(unsigned long bytes) { simt::lds_limit = bytes; }
extern "C" void simt_set_schedule(int mode, unsigned long seed) {
  simt::sched_mode = mode;
  simt::sched_state = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed * 0xD1B54A32D192ED03ull + 1);
}
