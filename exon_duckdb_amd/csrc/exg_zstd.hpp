// exg_zstd.hpp — Zstandard (RFC 8878) on the device: tables shared by the host walk over the frame / block headers
// (exg_zstd_index.cpp) and the kernels (exg_zstd.hip).
//
// Replaces the decompression the reference gets from DataFusion 28 `FileCompressionType::ZSTD` -> async-compression
// 0.4.0 -> zstd 0.12.3 (libzstd 1.5.2; rust/Cargo.lock:3875-3876), selected at rust/src/arrow_reader.rs:73 (".zst") and
// :87-88 (compression='zstd'); pinned by test_fastq_scan.test:22-32, 55-59 and test_fasta_scan.test:22-26, 45-49.
//
// Why the work is split the way it is: unlike DEFLATE, a zstd stream says where everything is.  Block_Size is in every
// 3-byte block header, the literals section states its own sizes, Number_of_Sequences is a byte or three — so the host
// finds every block of every frame by a pointer chase over a few bytes per block (no decoding), and the device can
//   (1) entropy-decode ALL blocks at once (Huffman literals, FSE sequences: one wavefront per block),
//   (2) turn block-local facts into global ones by a scan (output offsets, the repeat-offset history — whose update
//       rule composes: every slot becomes a constant or "an incoming slot minus k"),
//   (3) execute the sequences (literal runs + LZ77 copies), the only stage that depends on earlier OUTPUT.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace exg {
namespace zst {

static constexpr uint32_t kBlockMax = 128u << 10;  // Block_Maximum_Size (RFC 8878 3.1.1.2.4)
static constexpr uint32_t kNone = 0xFFFFFFFFu;
static constexpr uint64_t kWindowMax = 1ull << 27;  // libzstd's default ZSTD_d_windowLogMax: larger frames are refused

// status codes of a block / the whole decode (0 = ok)
enum : uint32_t {
    kOk = 0,
    kErrLiterals = 1,   // literals section inconsistent
    kErrHuffman = 2,    // bad Huffman tree description or stream
    kErrFse = 3,        // bad FSE table description
    kErrSequences = 4,  // sequence bitstream inconsistent
    kErrOffset = 5,     // a match reaches in front of the frame's content (or offset 0)
    kErrSize = 6,       // block or frame content size wrong
    kErrChecksum = 7,   // XXH64 of the content does not match the frame's checksum
};

// sequence offsets leave the FSE stage either resolved or "repeat-offset slot j of the block's incoming history, minus k"
static constexpr uint32_t kRepSym = 0x80000000u;  // | slot << 29 | k

struct Block {  // one entry per block, in file order; the host fills the first part from the headers
    uint64_t src_off;   // block content in the compressed buffer
    uint32_t src_size;  // Block_Size as stored (RLE: the regenerated size; its content is 1 byte)
    uint8_t type;       // 0 raw, 1 RLE, 2 compressed
    uint8_t first_of_frame;
    uint8_t lit_type;     // 0 raw, 1 RLE, 2 Huffman, 3 Huffman with the previous tree
    uint8_t lit_streams;  // 1 or 4
    uint32_t frame;
    uint32_t lit_hdr;    // bytes of the literals section header
    uint32_t lit_regen;  // Regenerated_Size
    uint32_t lit_csize;  // Compressed_Size (tree description + streams), types 2 / 3
    uint32_t nseq;
    uint32_t seq_hdr;     // offset (from the block's start) of the Symbol_Compression_Modes byte, if nseq > 0
    uint32_t huf_src;     // the block whose literals section carries the Huffman tree in use (kNone: none)
    uint32_t tbl_src[3];  // LL, OF, ML: the block whose sequences section defines the table in use (mode != Repeat)
    uint32_t pad0[2];
    uint64_t lit_off;  // this block's literals in the literal buffer
    uint64_t seq_off;  // this block's sequences in the sequence arrays
    // ---- written by the device ----
    uint32_t out_size;    // bytes the block regenerates (raw / RLE: set by the host)
    uint32_t status;      // kOk or the first error met
    uint32_t rep_out[3];  // repeat offsets at the block's end, symbolic in its incoming history (k_zst_sequences)
    uint32_t rep_in[3];   // repeat offsets at the block's start, resolved (k_zst_scan)
    uint64_t out_off;     // where the block's output begins (k_zst_scan), absolute in the output buffer
};
static_assert(sizeof(Block) == 120, "layout shared by host and device");

struct Frame {
    uint64_t content_size;  // Frame_Content_Size, ~0 when the header does not carry it
    uint64_t window;        // Window_Size
    uint32_t first_block, n_blocks;
    uint32_t has_checksum, checksum;  // Content_Checksum (low 32 bits of XXH64, seed 0)
    uint64_t out_off, out_size;       // filled from the block sizes
    uint64_t src_off;                 // where the frame begins in the file (its magic number)
};

// a run of consecutive blocks of one frame executed by one wavefront
struct Chunk {
    uint32_t first_block, n_blocks;
    uint64_t out_off;        // absolute position of the chunk's first byte
    uint64_t frame_out_off;  // ... of its frame's first byte
    uint64_t elem_off;       // where the chunk's elements go: byte chunks = out_off, symbol chunks = offset in d_sym
    uint32_t symbolic;       // 1: elements are 32-bit symbols (a byte, or "the byte d in front of the chunk": kSymRef | d)
    uint32_t pad;
    uint64_t size;           // bytes of output
    uint64_t byte_end;       // end of its frame's first chunk: the frame's positions below it are bytes from the start
};
static constexpr uint32_t kSymRef = 0x80000000u;

// a frame whose Content_Checksum the device did not verify (more than EXG_ZSTD_VERIFY_MAX bytes of content): verified on
// the host from a copy of the decoded bytes (host_verify)
struct PendingCheck {
    uint64_t out_off, size;
    uint32_t expect, frame;
};

struct Index {
    std::vector<Block> blocks;
    std::vector<Frame> frames;
    uint64_t lit_bytes = 0;  // literal buffer size (every block's literals padded to 16)
    uint64_t n_seq = 0;
    uint64_t known_out = 0;  // sum of the raw / RLE block sizes
    std::string error;       // set when the walk fails
    // when the walk fails inside a frame: that frame's header (its complete blocks are at blocks[open_frame.first_block ...])
    Frame open_frame;
    bool open_valid = false;
};

// Host: walk the frames and blocks of data[0, n).  false + idx.error on a malformed stream.
bool build_index(const uint8_t *data, uint64_t n, Index &idx);
// the same over file bytes [0, n) of fd, with small reads (nothing is mapped)
bool build_index_fd(int fd, uint64_t n, Index &idx);
// the index of the stream's first blocks (until their estimated output reaches stop_est): exg_zstd_index.cpp
bool build_index_prefix_fd(int fd, uint64_t n, uint64_t stop_est, Index &idx, bool *stopped);
// After a failed walk: keep what lies in front of the damage — the complete frames and the complete blocks of the frame that
// was open (as a frame without a checksum or a stated size) — so that a reader can hand out their rows before it reports
// idx.error, like a streaming decoder does with a truncated file.  false: nothing usable lies in front of it.
bool salvage_index(Index &idx);

// One ROUND of a stream: a run of consecutive blocks — whole frames, or a part of a frame — decoded on the device into a
// pooled block [front_reserve | history | produced | 64 zero bytes].  A stream of any size is decoded round by round with
// bounded memory (exg_rd_zstd.cpp), like a streaming decoder with its window: what a frame that goes on needs from earlier
// rounds is (a) the repeat offsets behind the last block (rep_in), (b) up to Window_Size bytes of its output (`d_history`:
// copied in front of this round's bytes, so that matches and the consumer's carried tail find them there), and (c) the
// compressed bytes of the blocks whose Huffman tree / FSE tables are repeated: they ride in front of the round's own blocks
// (`n_extra`: they are parsed again as table sources, not decoded; tbl_src / huf_src index THIS array).
struct RoundFrame {
    uint32_t first_block = 0, n_blocks = 0;  // in Round::blocks
    uint32_t frame_id = 0;                   // (messages)
    bool begins = true, ends = true;         // the round holds the frame's first / last block
    uint64_t history = 0;                    // !begins: bytes of this frame in front of the round's (<= Round::history)
    uint32_t has_checksum = 0, checksum = 0;
    // results: where its bytes lie (buffer coordinates: 0 = the first history byte), whether the device verified its checksum
    uint64_t out_off = 0, out_size = 0;
    bool verified = false;
};
struct Round {
    std::vector<Block> blocks;  // src_off relative to d_comp; out_size / status / rep_* / out_off come back filled
    uint32_t n_extra = 0;
    std::vector<RoundFrame> frames;
    uint32_t rep_in[3] = {1, 4, 8};
    const void *d_comp = nullptr;
    const void *d_history = nullptr;
    uint64_t history = 0;
    uint64_t front_reserve = 0;  // a multiple of 16
    uint64_t verify_max = 0;     // frames up to this many bytes that lie inside the round are hashed (XXH64) on the device
    uint64_t first_block_id = 0, comp_base = 0;  // (messages: the stream's index of blocks[n_extra], the file offset of d_comp)
    // results
    void *d_buf = nullptr;  // pooled block of `alloc` bytes; content at d_buf + front_reserve + history
    size_t alloc = 0;
    uint64_t produced = 0;
    uint32_t rep_out[3] = {1, 4, 8};
};
int decode_round(Round &R, void *stream);
// the same in three phases (exg_zstd.hip): begin (entropy stages, scan, chunk plan) fills R.rep_out, R.frames[].out_off /
// out_size and R.produced; enqueue (execution, resolve, checksums: launches only) needs R.d_history; wait hands R.d_buf over.
// A phase that fails has disposed of the context.  (Overlapping two rounds with them measured no gain: see exg_zstd.hip.)
struct RoundCtx;
int decode_round_begin(Round &R, void *stream, RoundCtx **ctx);
int decode_round_enqueue(Round &R, RoundCtx *ctx);  // = _exec + _resolve
// the enqueue phase in two halves: the chunks' execution needs nothing of the round in front (a frame that goes on writes
// symbols from its first chunk); the resolve launches need R.d_history
int decode_round_enqueue_exec(Round &R, RoundCtx *ctx);
int decode_round_enqueue_resolve(Round &R, RoundCtx *ctx);
int decode_round_wait(Round &R, RoundCtx *ctx);
void decode_round_abandon(RoundCtx *ctx);
uint64_t default_verify_max();

// exg_zstd_decode (include/exon_gpu.h) without the host half of the checksum verification: the frames left to it come back
// in *pending (NULL: they stay unverified).  host_verify: copies each such frame from d_out in pieces and hashes it (XXH64)
// on the calling thread, on a stream of its own; EXG_E_PARSE + *err ("Restored data doesn't match checksum ...") on a mismatch.
// front_reserve (a multiple of 16): bytes left free in front of the content in *d_out (the content begins at *d_out + front_reserve;
// the block was taken from the device pool with front_reserve + *produced + 64 bytes)
int decode(const uint8_t *h_comp, const void *d_comp, uint64_t n, void **d_out, uint64_t *produced, void *stream, std::vector<PendingCheck> *pending,
           Index *prebuilt = nullptr, uint64_t front_reserve = 0);  // prebuilt: build_index(h_comp, n) done by the caller (beside the upload)
int host_verify(const void *d_out, const std::vector<PendingCheck> &pending, int device, std::string *err);

}  // namespace zst
}  // namespace exg
