// region_inflate.hpp -- a stretch of ONE serial DEFLATE stream decoded sequentially on the host, from a known block boundary on.
//
// `fqtk demux` decodes serial gzip inputs on the device in chunks cut at block starts (include/fqtk_demux.h:
// fqtk_demuxer_stream_scan).  Some stretches of a valid file cannot be taken that way -- no dynamic-Huffman block starts to cut at
// (stored blocks, one block of many MB), a block that expands beyond any room for symbols -- and the reference reads every valid
// gzip file (/root/reference/src/bin/commands/demux.rs:844-849).  So such a stretch is decoded here, by fast_inflate.hpp's
// sequential decoder started at the bit the device's chain was verified up to, with the 32 KiB of text in front of it
// (fqtk_demuxer_stream_window), and its text is handed back (fqtk_demuxer_stream_commit_text).  It is also the second opinion
// before a stream is called corrupt: an error is reported only if this decoder fails at the same place.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "fast_inflate.hpp"

namespace fqtk_host {

class RegionInflate : public FastInflate {
  public:
    // data[0 .. n): the mapped file.
    void attach(const uint8_t *data, size_t n) { open(data, n, nullptr, /*with_output=*/true); }

    // Decodes whole blocks from bit `from_bit` (a block boundary) until the first block boundary at or behind `until_bit`, or the
    // member's final block, or -- at a block boundary -- `max_text` bytes of text; a single block larger than that is decoded whole.
    // window: the 32 KiB of text in front of from_bit (window[32767] = the byte right before it); nullptr: the member starts here.
    // window_valid: how many of them -- counted from the window's END -- the member really has in front of from_bit (a member that has
    // decoded less than 32 KiB so far: the rest of the window is filler, and a distance that reaches into it is "too far back" here as
    // in the streaming decoder, not a copy of zeros that fails the CRC much later: ADVICE r05).
    // text receives the bytes (appended); *end_bit the boundary reached, *final_block whether it ended the member;
    // window_after (32 KiB) the text in front of end_bit.  false: the stream is corrupt there (*err says how).
    bool run(uint64_t from_bit, const uint8_t *window, uint64_t until_bit, size_t max_text, std::vector<uint8_t> *text, uint64_t *end_bit, bool *final_block,
             uint8_t *window_after, std::string *err, size_t window_valid = kWindow) {
        uint8_t *base = obuf_.data();
        if (window) {
            const size_t valid = window_valid < kWindow ? window_valid : kWindow;
            std::memcpy(base, window + (kWindow - valid), valid);
            hist_ = valid;
        } else {
            hist_ = 0;
        }
        if (!seek_bit(from_bit, err)) return false;
        state_ = State::BlockStart;
        final_block_ = false;
        const size_t text0 = text->size();
        for (;;) {
            // one block, in pieces of the output buffer
            if (!start_block(err)) return false;
            while (state_ == State::Stored || state_ == State::Codes) {
                if (hist_ > kWindow) {   // slide: the last 32 KiB stay in front of the next piece
                    std::memmove(base, base + hist_ - kWindow, kWindow);
                    hist_ = kWindow;
                }
                uint8_t *const piece = base + hist_;
                uint8_t *op = piece;
                uint8_t *const stop = piece + kPiece;
                if (state_ == State::Stored) {
                    const size_t take_n = stored_left_ < (size_t)(stop - op) ? stored_left_ : (size_t)(stop - op);
                    if (in_end_ - ip_ < (ptrdiff_t)take_n) return fail(err, "stored block runs past the end of the file");
                    std::memcpy(op, ip_, take_n);
                    op += take_n;
                    ip_ += take_n;
                    stored_left_ -= take_n;
                    if (stored_left_ == 0) state_ = final_block_ ? State::Trailer : State::BlockStart;
                } else if (!decode(base, op, stop, err)) {
                    return false;
                }
                const size_t got = (size_t)(op - piece);
                text->insert(text->end(), piece, piece + got);
                hist_ += got;
            }
            // a block boundary
            if (final_block_ || bit_pos() >= until_bit || text->size() - text0 >= max_text) break;
        }
        *end_bit = bit_pos();
        *final_block = final_block_;
        if (window_after) {
            const size_t have = hist_ < kWindow ? hist_ : kWindow;
            std::memset(window_after, 0, kWindow - have);
            std::memcpy(window_after + kWindow - have, base + hist_ - have, have);
        }
        return true;
    }
};

}  // namespace fqtk_host
