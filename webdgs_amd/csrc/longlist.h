// Long tile lists: per-PIXEL walks for the blocks of tiles whose entry list exceeds a threshold.
//
// The rasterization kernels give one wave to an 8x8 pixel block, and that wave walks the tile's list record by record -- every record costs the
// block ~33 (forward) or ~113 (backward) wave-instructions however few of its 64 pixels the record touches.  A tile that holds thousands of small
// splats (distant detail collapsed onto one tile) is then ONE sequential chain as long as its list, while a pixel of it meets a few dozen records:
// 10 400 entries cost 0.34 + 0.86 ms, 40 000 cost 1.2 + 3.0 ms, next to a 0.63 ms step (profiles/r08l_long_list_scenes.txt).  The reference bounds
// the case by truncating the list at 8 192 entries (tiled-rasterizer.wgsl:59-60, 125); here it is bounded by giving every pixel its own list:
//
//   build   (sort.hip: segment_sort, which knows every tile's length)  tiles with more than `threshold` entries get four block records and
//           4 * ceil(n / 64) item slots; flags[tile] tells the main waves to leave those blocks alone -- if the frame's long tiles ALL found room
//           (ll_frame_on below: all or none);
//   count   per (block, chunk of 64 entries), in parallel: the chunk's records as the main kernel builds them (tile-level record table), and per
//           pixel the number of records whose extent box holds it;
//   scan    per block: per-pixel offsets of the chunks, rows needed = the longest per-pixel list; rows come out of a pool;
//   fill    per (block, chunk), in parallel: alpha of every (pixel, record inside its box), appended to the PIXEL's list: row j of a block holds
//           the j-th list element of each of its 64 pixels {alpha, colour, position in the tile list};
//   walk    per block, ONE wave: lane = pixel, trip j composites row j -- as many trips as the longest per-pixel list (until saturation), not as
//           the tile list; the backward kernel's helpers walk the same rows back to front, each lane adding its own contributions.
//
// None of this is a launch.  The tasks sit in one queue in the order above and are pulled, one atomic each, by the waves of the rasterization
// kernel itself as they finish their own blocks (the waves of the long tiles at once); a task waits only for tasks in front of it in the queue,
// which waves that are already running have taken -- so nothing can wait for a wave that has not started.  With no long tile in the frame a wave
// reads two header words (requested when it starts) and leaves: the path is always recorded and decided on the device.  Every operation keeps the operands it has
// in the wave-per-block walk and the parity oracle's own forms (raster.hip: EXACT), so results are bit-identical whichever path a block takes.
#pragma once
#include "common.h"
#include "dmath.h"

// header words (u32 hdr[LL_HDR_WORDS]; zeroed per frame by the scan kernel, counted up by segment_sort)
constexpr u32 LL_BLOCKS = 0u;       // block records wanted (4 per long tile); those below max_blocks exist
constexpr u32 LL_ITEMS = 1u;        // item slots wanted; a tile's slots exist if they end below max_items
constexpr u32 LL_FWD_HEAD = 2u;     // next task of the forward queue
constexpr u32 LL_BWD_HEAD = 3u;     // next task of the backward queue
constexpr u32 LL_ROWS = 4u;         // rows handed out
constexpr u32 LL_ROWS_WANTED = 5u;  // rows asked for (also by blocks that found the pool empty): what the host sizes the pool by
constexpr u32 LL_STALLED = 6u;      // != 0: a task gave up waiting for the tasks in front of it (never expected; the frame's long blocks are then not to be trusted)
constexpr u32 LL_HDR_WORDS = 8u;

constexpr u32 LL_NO_ROWS = 0xFFFFFFFFu;

struct LongBlock { u32 tile, sub, first_item, chunks; };
struct LongSync {
    u32 counted;    // count tasks of the block that have finished
    u32 row_base;   // written by the scan task: first row of the block's lists, or LL_NO_ROWS (pool empty, or lists as long as the tile list: plain walk)
    u32 scanned;    // 1 once row_base and the offsets are there
    u32 filled;     // fill tasks that have finished
    u32 rows;       // the longest per-pixel list
    u32 walked;     // 1: the forward walk has written the block's pixels through the lists (the backward helpers may use them)
    u32 pad0, pad1;
};

struct LongWork {
    u32* hdr;             // [LL_HDR_WORDS]
    u32* flags;           // [tiles]: bit b: block b of the tile is walked by tasks (forward); bit 4 + b: ... and its lists serve the backward pass
    LongBlock* blocks;    // [max_blocks]
    LongSync* sync;       // [max_blocks]
    u32* item_block;      // [max_items]: block record of an item slot
    u32* nlist;           // [max_items]: records the chunk kept
    u32* cnt;             // [max_items][64]: per pixel, records of the chunk whose box holds it
    u32* off;             // [max_items][64]: ... and where the chunk's part of the pixel's list starts
    u32* total;           // [max_blocks][64]: length of the pixel's list
    u32* jlast;           // [max_blocks][64]: list index of the pixel's last contributor + 1 (0: none) -- where the backward walk starts
    float4* records;      // [max_items][64][3]: geo, con, col of the chunk's kept records (per block)
    u32* rows;            // [max_rows][64][4]: per pixel {alpha, r | g (fp16 x 2), b (fp16), position in the tile list + 1}
    u32 max_blocks, max_items, max_rows, threshold;
    // (nullable) the forward pass's marks of tiles that hold a non-finite Splat (project.hip): such a tile is NOT given per-pixel lists -- what makes it
    // long is, in every run seen, a pile of NaN Gaussians behind a few real ones, which the EXACT body of raster.hip drops by the chunk
    const u32 *nf_stamp, *nf_frame;
};

// The primitives below are written WITHOUT lane-0 branches: every lane of the wave executes every operation (the one lane that counts adds 1, the
// others 0; uniform values are stored by all lanes).  A branch on the lane number inside the task loop, followed by a read of "the first lane", relies
// on the lanes having reconverged at that read -- which the compiler does not promise in a loop with many exits: the first version of this file hung
// there, lane 0 parked at a join the others never reached.
//
// Memory.  A task's results are read by tasks on other CUs and other XCDs of the same launch.  Made visible the textbook way -- plain stores, a release
// fence at agent scope, an acquire fence at the reader -- every task writes back and every reader invalidates a whole L2 (buffer_wbl2 / buffer_inv sc1:
// the XCDs' L2s are not coherent with each other for ordinary memory), which on a frame whose L2s are full of freshly written image data took longer
// than the work: 1.9 ms for the 1 400 tasks of the late regime's one long tile (profiles/r08o_*).  So everything that crosses between tasks of ONE launch
// is written and read with relaxed agent-scope atomic accesses of 32 bits (ll_st / ll_ld: sc1 loads and stores, coherent across the XCDs location by
// location, no cache maintenance), a writer waits for its stores to complete before it counts itself as done (ll_signal), and a reader does not
// load before it has seen the count (ll_wait).  What only the NEXT launch reads (records, jlast, the marks) is stored plainly.
WD_DEV void ll_st(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WD_DEV u32 ll_ld(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The wave waits until *p >= target (a counter of tasks in front of this one in the queue: their waves are running).  The wait is bounded -- a wave
// must be able to leave whatever happens: after ~0.1 s it notes the stall in the header (code, for the host to report) and returns false.
WD_DEV bool ll_wait(const u32* p, u32 target, u32* hdr, u32 code) {
    for (u32 spins = 0; spins < (1u << 19); spins++) {
        if ((u32)__builtin_amdgcn_readfirstlane((int)ll_ld(p)) >= target) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (orders the loads that follow behind this one; no cache maintenance)
            return true;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    atomicMax(&hdr[LL_STALLED], code);
    return false;
}
// The wave's ll_st stores have completed (s_waitcnt); then the task counts as done (one lane adds 1, the others 0).
WD_DEV void ll_signal(u32* p, u32 lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_fetch_add(p, lane == 0u ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The first of the next `batch` tasks of a queue (uniform), or 0xFFFFFFFF when it is exhausted.  A wave works its tasks off in order, so a task still
// waits only for tasks that running waves hold.
WD_DEV u32 ll_pull(u32* head, u32 n_tasks, u32 batch, u32 lane) {
    // (a look before the ticket: every wave of the launch comes here once its own block is done, and 32 000 read-modify-writes of ONE word take a
    // millisecond between them -- profiles/r08p_*; once the queue has run out the word is only read)
    if ((u32)__builtin_amdgcn_readfirstlane((int)ll_ld(head)) >= n_tasks) return 0xFFFFFFFFu;
    const u32 t = (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_fetch_add(head, lane == 0u ? batch : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // (lane 0: the value before its own addition)
    return t < n_tasks ? t : 0xFFFFFFFFu;
}
// Tickets cost ~28 ns each whoever draws them (one word, read-modify-written at the memory side: r08p).  A queue of a thousand tasks -- one long tile --
// is drawn one by one, so that every task has its own wave; longer queues are drawn in batches of up to 8: there are only ~2 000 waves to deal them to.
WD_DEV u32 ll_batch(u32 n_tasks) { return min(max(n_tasks >> 11, 1u), 8u); }
// The per-pixel lists are a remedy for a FEW long tiles in a frame: their lists are chains that one wave each walks while the rest of the chip has run
// out of work.  A frame FULL of long tiles (a dense cloud seen at a small viewport: the densify events' half-resolution metric views of config c3 want
// 81 000 chunk slots) keeps every CU busy with the wave-per-block walk, which does a third of the work the count / fill / walk tasks do between them:
// given room for all of it, such a view took 5.7 ms instead of 0.29 (profiles/r08s_event_timing.txt).  So the path is taken by ALL long tiles of a
// frame or by none: when the frame wants more block records or chunk slots than the scratch holds -- and the host does not let the scratch grow past a
// few thousand slots (Trainer: longLists.maxItemsCap) -- every tile stays with the main waves.
WD_DEV bool ll_frame_on(const LongWork& lw, u32 blocks_wanted, u32 items_wanted) { return blocks_wanted != 0u && blocks_wanted <= lw.max_blocks && items_wanted <= lw.max_items; }
