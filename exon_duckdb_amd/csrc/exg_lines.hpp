// exg_lines.hpp — launchers of the general line index (exg_lines.hip).
#pragma once
#include "exg_fastq_ws.hpp"

namespace exg {

// `gate`: device word; when non-null and zero at run time every kernel returns at once (used to
// run the general path only if the fused kernel raised `overflow`).
// Enqueue count -> scan -> emit on `stream`.  After it, hdr->total_lines lines are indexed in
// nl_pos (u64 offsets of each line's terminating '\n', virtual EOF terminators = n_bytes).
// line_flags (optional, lines_cap bytes): bit 0 = the line AFTER this one starts with '>', bit 1 = this
// line's '\n' is preceded by '\r'; not written for the virtual EOF terminators.
int launch_line_index(const uint8_t *d_in, uint64_t n_bytes, uint64_t lead, uint8_t *ws, const FastqWsLayout &l,
                      int eof_mode, uint64_t first_line_index, hipStream_t stream,
                      const unsigned int *gate = nullptr, uint8_t *line_flags = nullptr);

}  // namespace exg
