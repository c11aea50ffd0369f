// parallel_gunzip.hpp -- one gzip stream decoded by several threads.
//
// SURVEY 8(f) row 3 asks for a reader POOL; a BGZF file gives one for free (independent blocks, fastq_io.hpp), a plain
// `.fastq.gz` does not: it is one serial bit stream whose every match may reach 32 KiB back.  The way around it
// (the idea behind pugz / rapidgzip, rebuilt here on fast_inflate.hpp's tables):
//   * cut the next stretch of the compressed file into chunks; for every chunk but the first, SEARCH for a place where
//     a DEFLATE block starts (a bit offset whose dynamic-Huffman header parses into two complete codes -- a one-in-
//     many-millions accident otherwise);
//   * decode all chunks at once, each from its start to the next chunk's start.  A chunk does not know the 32 KiB
//     before it, so it decodes into 16-bit symbols: a byte, or "the byte at position j of the unknown window";
//     copies of such symbols copy the reference;
//   * afterwards the windows are handed down the line (the last 32 KiB of chunk k, resolved, are chunk k+1's) and
//     every chunk is resolved to bytes, again in parallel.
// Nothing here is trusted on probability: a chunk's output is used only if the chunk BEFORE it -- itself accepted,
// starting from the sequential decoder's exact position -- ended on exactly the bit this chunk started at, at a
// block boundary.  Then that bit IS a block start of the one true parse, and the chunk decoded what a sequential
// decoder would have.  Anything else (no start found, a false start, an error) discards the rest of the stretch and
// the sequential decoder carries on from the last accepted bit.  CRC32 and ISIZE of every member are still checked.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>   // crc32_combine

#include "env.hpp"

#include "fast_inflate.hpp"

namespace fqtk_host {

// One speculative chunk: DEFLATE blocks from a bit position, unknown window, into 16-bit symbols.
class SpecInflate : public FastInflate {
  public:
    struct Task {
        uint64_t start_bit = 0, stop_bit = 0;   // decode until the first block boundary at or after stop_bit
        std::vector<uint16_t> sym;              // < 256: the byte; else 256 + index into the 32 KiB before the chunk
        size_t n_sym = 0;
        uint64_t end_bit = 0;
        bool final_block = false, error = false;
        double seconds = 0;                     // (how long run() took)
    };
    // A chunk of highly compressible data could expand a thousandfold: past this many symbols it is given up (the
    // stretch then ends before it and the sequential decoder, which works in 4 MiB pieces, takes over for a while).
    static constexpr size_t kMaxSymbols = 48u << 20;

    // (a speculative decoder writes 16-bit symbols into its task, never into obuf_: no 4 MiB output buffer per worker)
    void attach(const uint8_t *data, size_t n) { open(data, n, nullptr, /*with_output=*/false); }

    // First bit position in [from, limit) where a non-final dynamic-Huffman block header parses; ~0 if none.
    uint64_t find_block_start(uint64_t from, uint64_t limit) {
        const uint64_t last = (uint64_t)(data_end_ - data_) * 8u;
        if (limit + 1024 > last) limit = last > 1024 ? last - 1024 : 0;   // a header needs room; the file's end is sequential anyway
        std::string scratch;
        for (uint64_t t = from; t < limit; ++t) {
            const uint64_t b = peek(t);
            // BFINAL = 0, BTYPE = 2 (binary 10, LSB first: bits 1-2 = 0, 1), HLIT <= 29, HDIST <= 29
            if ((b & 7u) != 4u) continue;
            if (((b >> 3) & 31u) > 29u || ((b >> 8) & 31u) > 29u) continue;
            const unsigned hclen = (unsigned)((b >> 13) & 15u) + 4u;
            // the code-length code must be complete: sum of 2^(7 - len) over its used symbols == 2^7
            unsigned kraft = 0;
            const uint64_t c0 = b >> 17, c1 = peek(t + 17 + 39);   // 3-bit fields 0..12 (b holds 57 bits) and 13..18
            for (unsigned k = 0; k < hclen; ++k) {
                const unsigned l = (unsigned)(((k < 13 ? c0 >> (3 * k) : c1 >> (3 * (k - 13)))) & 7u);
                if (l) kraft += 128u >> l;
            }
            if (kraft != 128u) continue;
            // the full header: both length sets decode and give complete codes
            if (!seek_bit(t, &scratch)) continue;
            state_ = State::BlockStart;
            if (start_block(&scratch) && state_ == State::Codes && !final_block_) return t;
        }
        return ~0ull;
    }

    void run(Task &tk) {
        const auto t0 = std::chrono::steady_clock::now();
        struct Stamp { Task &t; std::chrono::steady_clock::time_point t0; ~Stamp() { t.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } stamp{tk, t0};
        std::string err;
        tk.n_sym = 0;
        tk.error = tk.final_block = false;
        if (tk.sym.size() < (1u << 20)) tk.sym.resize(1u << 20);
        if (!seek_bit(tk.start_bit, &err)) { tk.error = true; return; }
        for (;;) {
            state_ = State::BlockStart;
            if (!start_block(&err)) { tk.error = true; return; }
            if (state_ == State::Stored) {
                if (in_end_ - ip_ < (ptrdiff_t)stored_left_ || tk.n_sym > kMaxSymbols) { tk.error = true; return; }
                reserve(tk, stored_left_);
                for (size_t i = 0; i < stored_left_; ++i) tk.sym[tk.n_sym + i] = ip_[i];
                tk.n_sym += stored_left_;
                ip_ += stored_left_;
                stored_left_ = 0;
            } else if (state_ == State::Codes) {
                if (!decode_symbols(tk)) { tk.error = true; return; }
            }
            if (final_block_) {
                tk.final_block = true;
                tk.end_bit = bit_pos();
                return;
            }
            const uint64_t at = bit_pos();
            if (at >= tk.stop_bit) { tk.end_bit = at; return; }
        }
    }

    // 16-bit symbols -> bytes, given the 32 KiB that preceded the chunk (window[32767] = the byte right before it).
    static void resolve(const uint16_t *sym, size_t n, const uint8_t *window, uint8_t *out) {
        uint8_t tab[256 + 32768];   // one look-up per symbol, no branch
        for (int b = 0; b < 256; ++b) tab[b] = (uint8_t)b;
        std::memcpy(tab + 256, window, 32768);
        size_t i = 0;
        for (; i + 4 <= n; i += 4) {
            out[i] = tab[sym[i]];
            out[i + 1] = tab[sym[i + 1]];
            out[i + 2] = tab[sym[i + 2]];
            out[i + 3] = tab[sym[i + 3]];
        }
        for (; i < n; ++i) out[i] = tab[sym[i]];
    }

  private:
    uint64_t peek(uint64_t bit) const {   // 57+ bits starting at `bit` (callers stay 1 KiB clear of the file's end)
        return load64(data_ + (bit >> 3)) >> (bit & 7u);
    }
    static void reserve(Task &tk, size_t more) {
        if (tk.n_sym + more + 600 > tk.sym.size()) tk.sym.resize(std::max(tk.sym.size() * 2, tk.n_sym + more + 600));
    }

    // One block's symbols (fast_inflate.hpp's loop, 16-bit output, references into the unknown window).
    bool decode_symbols(Task &tk) {
        uint64_t bb = bb_;
        unsigned bc = bc_;
        const uint8_t *ip = ip_;
        const uint32_t *const lit = lit_, *const dist = dist_;
        bool ok = true, eob = false;
        const uint8_t *in_limit = tail_active_ ? in_end_ + 8 : in_end_ - kTailAt;
        size_t o = tk.n_sym;
        uint16_t *out = tk.sym.data();
        size_t cap = tk.sym.size();
#define FQTK_REFILL() do { bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56; } while (0)
#define FQTK_DROP(n) do { bb >>= (n); bc -= (n); } while (0)
#define FQTK_PUT16(e) do { out[o] = (uint16_t)(((e) >> 16) & 0xFFu); out[o + 1] = (uint16_t)((e) >> 24); o += ((e) >> 8) & 3u; FQTK_DROP((e) & 0xFF); } while (0)
        while (!eob) {
            if (o + 600 > cap) {
                if (o > kMaxSymbols) { ok = false; break; }
                tk.n_sym = o;
                reserve(tk, 1u << 20);
                out = tk.sym.data();
                cap = tk.sym.size();
            }
            if (ip > in_limit) {
                if (tail_active_) { ok = false; break; }
                ip_ = ip;
                guard_tail();
                ip = ip_;
                in_limit = in_end_ + 8;
            }
            FQTK_REFILL();
            uint32_t e = lit[bb & ((1u << kLitBits) - 1)];
            if (e & kLit) {   // up to three look-ups of one or two literals on one refill, as in fast_inflate.hpp
                FQTK_PUT16(e);
                e = lit[bb & ((1u << kLitBits) - 1)];
                if (e & kLit) {
                    FQTK_PUT16(e);
                    e = lit[bb & ((1u << kLitBits) - 1)];
                    if (e & kLit) {
                        FQTK_PUT16(e);
                        continue;
                    }
                }
                FQTK_REFILL();
            }
            if (e & kSub) {
                FQTK_DROP(kLitBits);
                e = lit[(e >> 16) + (bb & ((1u << ((e >> 8) & 31u)) - 1))];
                if (e & kLit) {
                    FQTK_PUT16(e);
                    continue;
                }
            }
            if ((e & 0xFF) == 0) { ok = false; break; }
            FQTK_DROP(e & 0xFF);
            if (e & kEob) { eob = true; break; }
            const unsigned lx = (e >> 8) & 31u;
            const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1ull << lx) - 1));
            FQTK_DROP(lx);
            uint32_t d = dist[bb & ((1u << kDistBits) - 1)];
            if (d & kSub) {
                FQTK_DROP(kDistBits);
                d = dist[(d >> 16) + (bb & ((1u << ((d >> 8) & 31u)) - 1))];
            }
            if ((d & 0xFF) == 0) { ok = false; break; }
            FQTK_DROP(d & 0xFF);
            const unsigned dx = (d >> 8) & 31u;
            const uint32_t distance = (d >> 16) + (uint32_t)(bb & ((1ull << dx) - 1));
            FQTK_DROP(dx);
            if (distance > o + 32768u) { ok = false; break; }
            if (distance <= o) {
                const uint16_t *src = out + o - distance;
                if (distance >= 8) {   // eight symbols (16 bytes) at a time; may write up to 7 symbols past the match
                    uint16_t *dst = out + o, *const end = dst + len;
                    do {
                        std::memcpy(dst, src, 16);
                        dst += 8;
                        src += 8;
                    } while (dst < end);
                } else {
                    for (uint32_t i = 0; i < len; ++i) out[o + i] = src[i];
                }
            } else {
                // (part of) the source lies before the chunk: position 32768 + (o - distance) + i of the window
                const int64_t s0 = (int64_t)o - (int64_t)distance;
                for (uint32_t i = 0; i < len; ++i) {
                    const int64_t sp = s0 + i;
                    out[o + i] = sp >= 0 ? out[sp] : (uint16_t)(256 + 32768 + sp);
                }
            }
            o += len;
        }
#undef FQTK_REFILL
#undef FQTK_DROP
#undef FQTK_PUT16
        tk.n_sym = o;
        bb_ = bb;
        bc_ = bc;
        ip_ = ip;
        if (!ok || overrun()) return false;
        return true;
    }
};

class ParallelGunzip : public FastInflate {
  public:
    ~ParallelGunzip() {
        for (Stage &st : stages_)
            if (st.driver.joinable()) st.driver.join();
    }
    // threads: decoders working side by side (>= 2; 1 would be the sequential decoder with extra steps)
    // chunk: compressed bytes per decoder and stretch (2 MiB; tests use small ones to cross many boundaries)
    // head: bytes left free in front of every decoded chunk handed out by take_chunk() (room for the reader to put the
    // tail of the record that straddles two chunks)
    void open(const uint8_t *data, size_t n, CrcFn crc, unsigned threads, size_t chunk = kChunk, size_t head = 0) {
        FastInflate::open(data, n, crc);
        head_ = head;
        chunk_ = std::max<size_t>(chunk, 4096);
        stop_at_block_end_ = true;
        threads_ = std::max(2u, threads);
        for (Stage &st : stages_) {
            if (st.driver.joinable()) st.driver.join();
            st.launched = false;
            st.workers.resize(threads_);
            for (auto &w : st.workers) w.attach(data, n);
            st.tasks.assign(threads_, SpecInflate::Task{});
        }
        which_ = 0;
        resolved_.assign(threads_, ByteVec());
        chunk_crc_.assign(threads_, 0);
        emit_chunk_ = emit_off_ = n_ready_ = 0;
        rounds_ = fallbacks_ = cooldown_bit_ = 0;
    }

    bool next(const uint8_t **out, size_t *n, std::string *err) {
        for (;;) {
            if (emit_chunk_ < n_ready_) {   // pieces of the last stretch, in order
                const ByteVec &r = resolved_[emit_chunk_];
                const size_t have = r.size() - head_;
                const size_t take_n = std::min(kPiece, have - emit_off_);
                *out = reinterpret_cast<const uint8_t *>(r.data()) + head_ + emit_off_;
                *n = take_n;
                emit_off_ += take_n;
                if (emit_off_ == have) { ++emit_chunk_; emit_off_ = 0; }
                if (take_n) return true;
                continue;
            }
            if (state_ == State::Done) { *n = 0; return true; }
            if (state_ == State::BlockStart && worth_a_stretch(bit_pos())) {
                if (!stretch(err)) return false;
                if (n_ready_) continue;
            }
            // sequential: one block (or the header / trailer work between members)
            size_t got = 0;
            if (!FastInflate::next(out, &got, err)) return false;
            if (got) { *n = got; return true; }
            if (state_ == State::Done) { *n = 0; return true; }
        }
    }
    // A whole decoded chunk by swap instead of by copy, when one is next in line (out's old buffer is reused here).
    // (the chunk's bytes start at offset head of `out`)
    bool take_chunk(ByteVec &out) {
        if (emit_chunk_ >= n_ready_ || emit_off_ != 0 || resolved_[emit_chunk_].size() <= head_) return false;
        out.swap(resolved_[emit_chunk_]);
        ++emit_chunk_;
        return true;
    }
    uint64_t rounds() const { return rounds_; }
    uint64_t fallbacks() const { return fallbacks_; }
    const double *seconds() const { return seconds_; }   // waiting for the decoders, resolving, CRC

    static constexpr size_t kChunk = 2u << 20;   // compressed bytes per chunk

  private:
    // One stretch in flight: where its chunks start and what they decoded.  Two of them take turns, so that the
    // decoders of the NEXT stretch already run (from the bit this one ended on, known as soon as its chunks have been
    // checked) while this one is resolved, checksummed and handed out.
    struct Stage {
        std::vector<SpecInflate> workers;
        std::vector<SpecInflate::Task> tasks;
        std::vector<uint64_t> starts;
        std::vector<unsigned> order;   // chunks that have a start, in file order
        uint64_t p0 = 0;
        bool launched = false;
        std::thread driver;            // finds the starts, then runs the decoders, and ends
    };

    bool worth_a_stretch(uint64_t pos) const {
        return pos >= cooldown_bit_ && (size_t)(data_end_ - data_) - (size_t)(pos >> 3) > 3 * chunk_;
    }

    void launch(Stage &st, uint64_t p0) {
        st.p0 = p0;
        st.launched = true;
        st.driver = std::thread([this, &st, p0] {
            const uint64_t file_bits = (uint64_t)(data_end_ - data_) * 8u;
            const unsigned K = threads_;
            st.starts.assign(K, ~0ull);
            st.starts[0] = p0;
            {   // where the other chunks can start
                std::vector<std::thread> th;
                for (unsigned k = 1; k < K; ++k)
                    th.emplace_back([&, k] {
                        const uint64_t from = ((p0 >> 3) + (uint64_t)k * chunk_) * 8u;
                        if (from + 8 * chunk_ / 2 < file_bits) st.starts[k] = st.workers[k].find_block_start(from, from + 8 * chunk_ / 2);
                    });
                for (auto &t : th) t.join();
            }
            st.order.clear();
            for (unsigned k = 0; k < K; ++k)
                if (st.starts[k] != ~0ull) st.order.push_back(k);
            const uint64_t stretch_end = std::min<uint64_t>(file_bits, ((p0 >> 3) + (uint64_t)K * chunk_) * 8u);
            std::vector<std::thread> th;
            for (size_t j = 0; j < st.order.size(); ++j) {
                SpecInflate::Task &tk = st.tasks[st.order[j]];
                tk.start_bit = st.starts[st.order[j]];
                tk.stop_bit = j + 1 < st.order.size() ? st.starts[st.order[j + 1]] : stretch_end;
                th.emplace_back([&st, j] { st.workers[st.order[j]].run(st.tasks[st.order[j]]); });
            }
            for (auto &t : th) t.join();
        });
    }

    // Decodes the next threads_ chunks side by side; leaves their bytes in resolved_[0 .. n_ready_) and this
    // (sequential) decoder positioned behind the last accepted one.  n_ready_ == 0: nothing accepted.
    bool stretch(std::string *err) {
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_mark = now();
        auto lap = [&](int k) { const double t = now(); seconds_[k] += t - t_mark; t_mark = t; };
        ++rounds_;
        emit_chunk_ = emit_off_ = n_ready_ = 0;
        const uint64_t p0 = bit_pos();
        Stage &st = stages_[which_];
        if (st.launched && st.p0 != p0) {   // started for a position we did not arrive at (a sequential interlude): void
            st.driver.join();
            st.launched = false;
        }
        if (!st.launched) launch(st, p0);
        st.driver.join();
        st.launched = false;
        const std::vector<unsigned> &order = st.order;
        const std::vector<SpecInflate::Task> &tasks = st.tasks;
        if (env_on("FQTK_PG_DEBUG"))
            for (size_t j = 0; j < order.size(); ++j)
                std::fprintf(stderr, "chunk %zu: %.1f ms, %zu symbols, error %d\n", j, tasks[order[j]].seconds * 1e3, tasks[order[j]].n_sym, (int)tasks[order[j]].error);
        lap(0);
        // the chain of trust: chunk j+1 counts only if chunk j ended exactly where it starts
        size_t accepted = 0;
        for (size_t j = 0; j < order.size(); ++j) {
            const SpecInflate::Task &tk = tasks[order[j]];
            if (tk.error) break;
            ++accepted;
            if (tk.final_block || j + 1 == order.size() || tk.end_bit != st.starts[order[j + 1]]) break;
        }
        if (accepted < order.size()) {
            // something did not line up (no start where one was believed, a chunk that expands beyond reason, damage):
            // sequential for the next few chunks' worth before speculating again
            ++fallbacks_;
            cooldown_bit_ = p0 + 8ull * 4 * chunk_;
        }
        if (accepted == 0) return true;   // the sequential decoder goes on from here (and reports damage, if that is what it was)
        const SpecInflate::Task &last = tasks[order[accepted - 1]];
        // the next stretch starts decoding now, from the bit this one ended on
        if (accepted == order.size() && !last.final_block && worth_a_stretch(last.end_bit)) launch(stages_[which_ ^ 1], last.end_bit);
        // windows down the line, then every chunk to bytes
        std::vector<std::vector<uint8_t>> windows(accepted + 1, std::vector<uint8_t>(32768, 0));
        {
            const size_t have = std::min<size_t>(hist_, 32768);
            std::memcpy(windows[0].data() + 32768 - have, obuf_.data() + hist_ - have, have);
        }
        for (size_t j = 0; j < accepted; ++j) {   // the last 32 KiB of chunk j, resolved = the window of chunk j + 1
            const SpecInflate::Task &tk = tasks[order[j]];
            const size_t tail = std::min<size_t>(tk.n_sym, 32768);
            std::vector<uint8_t> &w = windows[j + 1];
            if (tail < 32768) std::memcpy(w.data(), windows[j].data() + tail, 32768 - tail);
            SpecInflate::resolve(tk.sym.data() + tk.n_sym - tail, tail, windows[j].data(), w.data() + 32768 - tail);
        }
        {
            std::vector<std::thread> th;
            for (size_t j = 0; j < accepted; ++j)
                th.emplace_back([&, j] {
                    const SpecInflate::Task &tk = tasks[order[j]];
                    resolved_[j].resize(head_ + tk.n_sym);
                    SpecInflate::resolve(tk.sym.data(), tk.n_sym, windows[j].data(), reinterpret_cast<uint8_t *>(resolved_[j].data()) + head_);
                    chunk_crc_[j] = crc_fn_(0, resolved_[j].data() + head_, tk.n_sym);   // folded into the member's CRC below
                });
            for (auto &t : th) t.join();
        }
        lap(1);
        for (size_t j = 0; j < accepted; ++j) {
            crc_ = (uint32_t)crc32_combine(crc_, chunk_crc_[j], (z_off_t)(resolved_[j].size() - head_));
            isize_ += (uint32_t)(resolved_[j].size() - head_);
        }
        lap(2);
        n_ready_ = accepted;
        // carry on behind the last accepted chunk, with its window as history
        std::memcpy(obuf_.data(), windows[accepted].data(), 32768);
        hist_ = 32768;
        if (!seek_bit(last.end_bit, err)) return false;
        final_block_ = last.final_block;
        state_ = last.final_block ? State::Trailer : State::BlockStart;
        which_ ^= 1;
        return true;
    }

    unsigned threads_ = 2;
    size_t chunk_ = kChunk;
    Stage stages_[2];
    int which_ = 0;
    std::vector<ByteVec> resolved_;
    size_t head_ = 0;
    std::vector<uint32_t> chunk_crc_;
    size_t emit_chunk_ = 0, emit_off_ = 0, n_ready_ = 0;
    uint64_t rounds_ = 0, fallbacks_ = 0, cooldown_bit_ = 0;
    double seconds_[4] = {0, 0, 0, 0};
};

}  // namespace fqtk_host
