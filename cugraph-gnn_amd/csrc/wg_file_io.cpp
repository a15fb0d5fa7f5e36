// Binary file <-> DISTRIBUTED handle (include/wgamd_comm.h: wholememory_load_from_file / wholememory_store_to_file).
//
// File format and sharding rules are the reference's (/root/reference/cpp/src/wholememory/file_io.cpp:1893-2160,
// cpp/include/wholememory/wholememory.h:422-461): a file is a headerless array of `file_entry_size`-byte entries, the
// files of a list are read as one concatenated array, and entry e of that array lands in the handle's entry e
// (round_robin_size == 0) or, with round-robin sharding, in rank (e / rr) % W at local entry (e / rr / W) * rr + e % rr.
// In memory an entry occupies `memory_entry_size` bytes (the row stride) and the payload starts `memory_offset` bytes in.
// Own design: every rank turns its share into a list of contiguous entry runs, each run is pread() into a pinned
// staging buffer and moved with one strided hipMemcpy2DAsync while the next chunk is being read (two buffers).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <vector>

#include "wg_common.hpp"
#include "wgamd_comm.h"

namespace {

using namespace wgamd;

constexpr size_t kStageBytes = 16u << 20;

struct pinned_pair {
  char* buf[2]       = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  hipStream_t stream = nullptr;
  size_t bytes;
  explicit pinned_pair(size_t n) : bytes(n)
  {
    for (int i = 0; i < 2; i++) {
      WG_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&buf[i]), n, hipHostMallocDefault));
      WG_HIP_CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    WG_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  }
  ~pinned_pair()
  {
    if (stream) {
      (void)hipStreamSynchronize(stream);
      (void)hipStreamDestroy(stream);
    }
    for (int i = 0; i < 2; i++) {
      if (done[i]) (void)hipEventDestroy(done[i]);
      if (buf[i]) (void)hipHostFree(buf[i]);
    }
  }
};

struct file_set {
  std::vector<int> fds;
  std::vector<size_t> first_entry;  // [n+1] prefix sum of entries per file
  ~file_set()
  {
    for (int fd : fds)
      if (fd >= 0) close(fd);
  }
};

void read_fully(int fd, char* dst, size_t bytes, off_t offset, const char* name)
{
  while (bytes > 0) {
    ssize_t r = pread(fd, dst, bytes, offset);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) throw logic_error(fmt("reading %s at offset %lld failed: %s", name, (long long)offset,
                                      r == 0 ? "unexpected end of file" : strerror(errno)));
    dst += r;
    bytes -= (size_t)r;
    offset += r;
  }
}

struct run {
  size_t file_entry, local_entry, count;
};

}  // namespace

extern "C" {

wholememory_error_code_t wholememory_load_from_file(wholememory_handle_t handle, size_t memory_offset,
                                                    size_t memory_entry_size, size_t file_entry_size,
                                                    const char** file_names, int file_count, int round_robin_size)
{
  return guarded("wholememory_load_from_file", [&] {
    WG_REQUIRE_INPUT(handle && file_names && file_count > 0, "null handle / empty file list");
    WG_REQUIRE_INPUT(file_entry_size > 0 && memory_offset + file_entry_size <= memory_entry_size,
                     "entry of %zu B at offset %zu does not fit the memory entry stride %zu", file_entry_size,
                     memory_offset, memory_entry_size);
    WG_REQUIRE_INPUT(round_robin_size >= 0, "round_robin_size must be >= 0");
    // the copies below run on a private stream: order them after everything already queued on this device (a fill or a
    // scatter into the same memory on the caller's stream), as the reference's synchronous cudaMemcpy does implicitly
    WG_HIP_CHECK(hipDeviceSynchronize());
    const size_t gran = wholememory_get_data_granularity(handle);
    WG_REQUIRE_INPUT(gran % memory_entry_size == 0, "memory entry stride %zu does not divide the handle granularity %zu",
                     memory_entry_size, gran);
    wholememory_comm_t comm = nullptr;
    WG_EXPECTS(wholememory_get_communicator(&comm, handle) == WHOLEMEMORY_SUCCESS, "no communicator");
    int rank = 0, W = 1;
    wholememory_communicator_get_rank(&rank, comm);
    wholememory_communicator_get_size(&W, comm);
    char* local_ptr = nullptr;
    size_t local_size = 0, local_offset = 0;
    WG_EXPECTS(wholememory_get_local_memory(reinterpret_cast<void**>(&local_ptr), &local_size, &local_offset, handle) ==
                 WHOLEMEMORY_SUCCESS, "no local memory");
    const size_t local_entries = local_size / memory_entry_size, local_first = local_offset / memory_entry_size;
    const size_t total_entries = wholememory_get_total_size(handle) / memory_entry_size;

    file_set files;
    files.first_entry.push_back(0);
    for (int i = 0; i < file_count; i++) {
      struct stat st;
      if (stat(file_names[i], &st) != 0) throw invalid_input(fmt("file %s: %s", file_names[i], strerror(errno)));
      if ((size_t)st.st_size % file_entry_size != 0)
        throw invalid_input(fmt("file %s: size %lld is not a multiple of the entry size %zu", file_names[i],
                                (long long)st.st_size, file_entry_size));
      files.first_entry.push_back(files.first_entry.back() + (size_t)st.st_size / file_entry_size);
      int fd = open(file_names[i], O_RDONLY);
      if (fd < 0) throw invalid_input(fmt("open %s for read failed: %s", file_names[i], strerror(errno)));
      files.fds.push_back(fd);
    }
    const size_t file_entries = files.first_entry.back();
    WG_REQUIRE_INPUT(file_entries <= total_entries, "the files hold %zu entries, the WholeMemory only %zu", file_entries,
                     total_entries);

    // ---- my share as contiguous runs --------------------------------------------------------------
    std::vector<run> runs;
    if (round_robin_size == 0) {
      size_t lo = std::min(local_first, file_entries), hi = std::min(local_first + local_entries, file_entries);
      if (hi > lo) runs.push_back(run{lo, 0, hi - lo});
    } else {
      const size_t rr = (size_t)round_robin_size;
      WG_REQUIRE_INPUT(rr <= file_entries / (size_t)W || file_entries == 0, "illegal round_robin_size");
      for (size_t k = 0;; k++) {
        size_t g = (k * (size_t)W + (size_t)rank) * rr;
        if (g >= file_entries) break;
        size_t cnt = std::min(rr, file_entries - g);
        WG_REQUIRE_INPUT(k * rr + cnt <= local_entries, "round-robin shard of rank %d does not fit its %zu local entries",
                         rank, local_entries);
        runs.push_back(run{g, k * rr, cnt});
      }
    }

    // ---- pread -> pinned -> strided copy, double-buffered ---------------------------------------------
    const size_t chunk_entries = std::max<size_t>(1, kStageBytes / file_entry_size);
    pinned_pair stage(chunk_entries * file_entry_size);
    int which = 0;
    bool used[2] = {false, false};
    for (const run& r : runs) {
      size_t done = 0;
      while (done < r.count) {
        const size_t n = std::min(chunk_entries, r.count - done);
        if (used[which]) WG_HIP_CHECK(hipEventSynchronize(stage.done[which]));
        // the chunk may straddle file boundaries
        size_t e = r.file_entry + done, filled = 0;
        while (filled < n) {
          size_t f   = (size_t)(std::upper_bound(files.first_entry.begin(), files.first_entry.end(), e) -
                              files.first_entry.begin()) - 1;
          size_t can = std::min(n - filled, files.first_entry[f + 1] - e);
          read_fully(files.fds[f], stage.buf[which] + filled * file_entry_size, can * file_entry_size,
                     (off_t)((e - files.first_entry[f]) * file_entry_size), file_names[f]);
          filled += can;
          e += can;
        }
        char* dst = local_ptr + (r.local_entry + done) * memory_entry_size + memory_offset;
        WG_HIP_CHECK(hipMemcpy2DAsync(dst, memory_entry_size, stage.buf[which], file_entry_size, file_entry_size, n,
                                      hipMemcpyDefault, stage.stream));
        WG_HIP_CHECK(hipEventRecord(stage.done[which], stage.stream));
        used[which] = true;
        which ^= 1;
        done += n;
      }
    }
    WG_HIP_CHECK(hipStreamSynchronize(stage.stream));
    if (wholememory_communicator_barrier(comm) != WHOLEMEMORY_SUCCESS) throw comm_error("barrier failed");
  });
}

wholememory_error_code_t wholememory_store_to_file(wholememory_handle_t handle, size_t memory_offset,
                                                   size_t memory_entry_stride, size_t file_entry_size,
                                                   const char* local_file_name)
{
  return guarded("wholememory_store_to_file", [&] {
    WG_REQUIRE_INPUT(handle && local_file_name, "null handle / file name");
    WG_REQUIRE_INPUT(file_entry_size > 0 && memory_offset + file_entry_size <= memory_entry_stride,
                     "entry of %zu B at offset %zu does not fit the memory entry stride %zu", file_entry_size,
                     memory_offset, memory_entry_stride);
    const size_t gran = wholememory_get_data_granularity(handle);
    WG_REQUIRE_INPUT(gran % memory_entry_stride == 0, "memory entry stride %zu does not divide the granularity %zu",
                     memory_entry_stride, gran);
    wholememory_comm_t comm = nullptr;
    WG_EXPECTS(wholememory_get_communicator(&comm, handle) == WHOLEMEMORY_SUCCESS, "no communicator");
    WG_HIP_CHECK(hipDeviceSynchronize());  // pending writes of any stream into the handle's memory must be visible
    if (wholememory_communicator_barrier(comm) != WHOLEMEMORY_SUCCESS) throw comm_error("barrier failed");
    char* local_ptr = nullptr;
    size_t local_size = 0, local_offset = 0;
    WG_EXPECTS(wholememory_get_local_memory(reinterpret_cast<void**>(&local_ptr), &local_size, &local_offset, handle) ==
                 WHOLEMEMORY_SUCCESS, "no local memory");
    const size_t entries = local_size / memory_entry_stride;
    int fd = open(local_file_name, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw invalid_input(fmt("open %s for write failed: %s", local_file_name, strerror(errno)));
    try {
      const size_t chunk_entries = std::max<size_t>(1, kStageBytes / file_entry_size);
      pinned_pair stage(chunk_entries * file_entry_size);
      // device -> pinned of chunk c+1 overlaps the write() of chunk c
      size_t issued = 0, written = 0;
      size_t n_in[2] = {0, 0};
      auto issue = [&](int b) {
        n_in[b] = std::min(chunk_entries, entries - issued);
        if (n_in[b] == 0) return;
        WG_HIP_CHECK(hipMemcpy2DAsync(stage.buf[b], file_entry_size, local_ptr + issued * memory_entry_stride + memory_offset,
                                      memory_entry_stride, file_entry_size, n_in[b], hipMemcpyDefault, stage.stream));
        WG_HIP_CHECK(hipEventRecord(stage.done[b], stage.stream));
        issued += n_in[b];
      };
      issue(0);
      for (int b = 0; written < entries; b ^= 1) {
        issue(b ^ 1);
        WG_HIP_CHECK(hipEventSynchronize(stage.done[b]));
        const char* src = stage.buf[b];
        size_t bytes    = n_in[b] * file_entry_size;
        while (bytes > 0) {
          ssize_t w = write(fd, src, bytes);
          if (w < 0 && errno == EINTR) continue;
          if (w <= 0) throw logic_error(fmt("writing %s failed: %s", local_file_name, strerror(errno)));
          src += w;
          bytes -= (size_t)w;
        }
        written += n_in[b];
      }
    } catch (...) {
      close(fd);
      throw;
    }
    if (close(fd) != 0) throw logic_error(fmt("closing %s failed: %s", local_file_name, strerror(errno)));
    if (wholememory_communicator_barrier(comm) != WHOLEMEMORY_SUCCESS) throw comm_error("barrier failed");
  });
}

}  // extern "C"
