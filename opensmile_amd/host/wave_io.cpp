// RIFF/WAVE ingest: the header walk of smilePcm_readWaveHeader (smileUtil.c:2374-2487).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>

#include "smilehip_host.hpp"

namespace smilehip_host {

namespace {
struct ChunkHead { uint32_t id, size; };
uint32_t le32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t le16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
bool read_chunk_head(FILE *f, ChunkHead &h) {
  unsigned char b[8];
  if (fread(b, 1, 8, f) != 8) return false;
  h.id = le32(b);
  h.size = le32(b + 4);
  return true;
}
}  // namespace

namespace {
// the header walk; on success the stream stands at the first sample and info.n_blocks is what the file really holds
bool parse_wave_header(FILE *f, const std::string &path, WaveInfo &info, std::string &err) {
  auto fail = [&](const std::string &m) { err = m + " ('" + path + "')"; return false; };
  unsigned char head[12];
  if (fread(head, 1, 12, f) != 12) return fail("file too short for a RIFF header");
  if (le32(head) != 0x46464952u || le32(head + 8) != 0x45564157u) return fail("bogus wave/riff header");
  ChunkHead ch;
  if (!read_chunk_head(f, ch)) return fail("file ends inside the sub-chunk header");
  while (ch.id != 0x20746D66u) {                       // "fmt " need not be the first sub-chunk
    fseek(f, (long)ch.size + (long)(ch.size % 2), SEEK_CUR);
    if (!read_chunk_head(f, ch)) return fail("no fmt chunk");
  }
  if (ch.size != 16 && ch.size != 18 && ch.size != 40) return fail("fmt chunk of unsupported size");
  unsigned char fmt[16];
  if (fread(fmt, 1, 16, f) != 16) return fail("file ends inside the fmt chunk");
  if (ch.size > 16) fseek(f, (long)ch.size - 16, SEEK_CUR);
  const uint16_t audio_format = le16(fmt), n_chan = le16(fmt + 2);
  const uint32_t rate = le32(fmt + 4);
  const uint16_t block_align = le16(fmt + 12), bits = le16(fmt + 14);
  if (audio_format != 1 && audio_format != 3) return fail("only PCM and IEEE float wave formats are supported");
  if (n_chan == 0 || block_align == 0) return fail("fmt chunk with zero channels or block size");
  if (!read_chunk_head(f, ch)) return fail("no data chunk");
  while (ch.id != 0x61746164u) {
    fseek(f, (long)ch.size + (long)(ch.size % 2), SEEK_CUR);
    if (!read_chunk_head(f, ch)) return fail("no data chunk");
  }
  info.sample_type = audio_format;
  info.sample_rate = (long)rate;
  info.n_chan = n_chan;
  info.n_bps = block_align / n_chan;
  info.n_bits = bits;
  info.n_blocks = (long)(ch.size / block_align);
  info.block_size = block_align;
  info.header_offset = ftell(f);
  // the header's data size is not trusted before allocating: files written to a pipe carry 0 or 0xFFFFFFFF there
  // (the reference's cWaveSource reads block by block until EOF): clamp to what the file really holds
  {
    const long here = ftell(f);
    long remaining = -1;
    if (here >= 0 && fseek(f, 0, SEEK_END) == 0) {
      const long end = ftell(f);
      if (end >= here) remaining = end - here;
      fseek(f, here, SEEK_SET);
    }
    bool to_eof = remaining >= 0 && (uint64_t)ch.size > (uint64_t)remaining;
    if (remaining >= 8 && ch.size == 0) {
      // size 0: a stream written to a pipe (audio until EOF) -- unless what follows is a chain of well-formed chunks (an empty
      // data chunk before LIST / id3 / ... metadata) that walks EXACTLY to the end of the file: four printable id bytes, a size,
      // the (word-aligned) payload, again. Samples that merely look like one chunk header do not pass that walk.
      long pos = here;
      const long end = here + remaining;
      bool chain = true;
      int n_chunks = 0;
      while (chain && pos < end) {
        unsigned char h[8];
        if (end - pos < 8 || fseek(f, pos, SEEK_SET) != 0 || fread(h, 1, 8, f) != 8) { chain = false; break; }
        for (int q = 0; q < 4 && chain; ++q)
          chain = (h[q] >= 'A' && h[q] <= 'Z') || (h[q] >= 'a' && h[q] <= 'z') || (h[q] >= '0' && h[q] <= '9') || h[q] == ' ';
        const uint64_t sz = (uint64_t)h[4] | ((uint64_t)h[5] << 8) | ((uint64_t)h[6] << 16) | ((uint64_t)h[7] << 24);
        const uint64_t adv = 8 + sz + ((sz & 1) && (uint64_t)pos + 8 + sz < (uint64_t)end ? 1 : 0);     // RIFF pads odd chunks
        if (!chain || adv > (uint64_t)(end - pos)) { chain = false; break; }
        pos += (long)adv;
        ++n_chunks;
      }
      fseek(f, here, SEEK_SET);
      to_eof = !(chain && n_chunks > 0 && pos == end);
    } else if (remaining >= 0 && remaining < 8 && ch.size == 0) {
      to_eof = remaining > 0;
    }
    if (to_eof) info.n_blocks = remaining / block_align;
    else if (remaining >= 0 && (uint64_t)info.n_blocks * block_align > (uint64_t)remaining) info.n_blocks = remaining / block_align;
  }
  return true;
}
}  // namespace

namespace {
// The common layout -- "fmt " and "data" headers inside the first 4 KB, a data size the file really holds -- from ONE pread and
// one fstat (a corpus of 10^5 files is 10^5 header walks: fopen / fseek / ftell are several system calls and a heap buffer
// each). Returns 1: parsed, 0: take the general walk (parse_wave_header: the same answer, every case), -1: I/O error.
int parse_wave_header_mem(int fd, WaveInfo &info) {
  unsigned char b[4096];
  const ssize_t n = pread(fd, b, sizeof(b), 0);
  if (n < 0) return -1;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return 0;
  if (n < 12 || le32(b) != 0x46464952u || le32(b + 8) != 0x45564157u) return 0;
  size_t pos = 12;
  auto head = [&](uint32_t &id, uint32_t &size) {
    if (pos + 8 > (size_t)n) return false;
    id = le32(b + pos); size = le32(b + pos + 4);
    pos += 8;
    return true;
  };
  uint32_t id, size;
  if (!head(id, size)) return 0;
  while (id != 0x20746D66u) {
    pos += (size_t)size + (size % 2);
    if (!head(id, size)) return 0;
  }
  if ((size != 16 && size != 18 && size != 40) || pos + size > (size_t)n) return 0;
  const unsigned char *fmt = b + pos;
  pos += size;
  const uint16_t audio_format = le16(fmt), n_chan = le16(fmt + 2);
  const uint32_t rate = le32(fmt + 4);
  const uint16_t block_align = le16(fmt + 12), bits = le16(fmt + 14);
  if ((audio_format != 1 && audio_format != 3) || n_chan == 0 || block_align == 0) return 0;     // (the general walk words the error)
  if (!head(id, size)) return 0;
  while (id != 0x61746164u) {
    pos += (size_t)size + (size % 2);
    if (!head(id, size)) return 0;
  }
  const int64_t remaining = (int64_t)st.st_size - (int64_t)pos;
  if (remaining < 0 || size == 0 || (uint64_t)size > (uint64_t)remaining) return 0;   // pipe-written sizes, truncated files: the general walk
  info.sample_type = audio_format;
  info.sample_rate = (long)rate;
  info.n_chan = n_chan;
  info.n_bps = block_align / n_chan;
  info.n_bits = bits;
  info.n_blocks = (long)(size / block_align);
  info.block_size = block_align;
  info.header_offset = (long)pos;
  return 1;
}
}  // namespace

bool probe_wave_file(const std::string &path, WaveInfo &info, std::string &err) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) { err = "cannot open '" + path + "'"; return false; }
  const int quick = parse_wave_header_mem(fd, info);
  close(fd);
  if (quick == 1) return true;
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open '" + path + "'"; return false; }
  const bool ok = parse_wave_header(f, path, info, err);
  fclose(f);
  return ok;
}

long read_wave_data(const std::string &path, const WaveInfo &info, void *dst, size_t bytes, std::string &err) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) { err = "cannot open '" + path + "'"; return -1; }
  size_t got = 0;
  while (got < bytes) {
    const ssize_t k = pread(fd, static_cast<char *>(dst) + got, bytes - got, (off_t)info.header_offset + (off_t)got);
    if (k < 0) { err = "error reading '" + path + "'"; close(fd); return -1; }
    if (k == 0) break;                                   // (a file that shrank since the probe: cWaveSource would stop here too)
    got += (size_t)k;
  }
  close(fd);
  return (long)(got / (size_t)info.block_size);
}

bool read_wave_file(const std::string &path, WaveInfo &info, std::vector<unsigned char> &data, std::string &err) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open '" + path + "'"; return false; }
  if (!parse_wave_header(f, path, info, err)) { fclose(f); return false; }
  const int block_align = info.block_size;
  data.resize((size_t)info.n_blocks * block_align);
  const size_t got = data.empty() ? 0 : fread(data.data(), 1, data.size(), f);
  if (got < data.size()) {                               // cWaveSource reads until EOF: a short data chunk just ends earlier
    info.n_blocks = (long)(got / block_align);
    data.resize((size_t)info.n_blocks * block_align);
  }
  fclose(f);
  return true;
}

}  // namespace smilehip_host
