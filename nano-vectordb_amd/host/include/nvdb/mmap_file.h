// Read-only memory mapping of a whole file (RAII).  Surface of reference include/nvdb/mmap_file.h:9-33.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace nvdb {

class MmapFile {
 public:
  MmapFile() = default;
  ~MmapFile() { reset(); }
  MmapFile(const MmapFile&) = delete;
  MmapFile& operator=(const MmapFile&) = delete;
  MmapFile(MmapFile&& o) noexcept { steal(o); }
  MmapFile& operator=(MmapFile&& o) noexcept { if (this != &o) { reset(); steal(o); } return *this; }

  void open_readonly(const std::string& path);   // throws std::runtime_error
  const uint8_t* data() const { return base_; }
  size_t size() const { return len_; }
  bool is_open() const { return base_ != nullptr; }

 private:
  void reset();
  void steal(MmapFile& o) { base_ = o.base_; len_ = o.len_; fd_ = o.fd_; o.base_ = nullptr; o.len_ = 0; o.fd_ = -1; }
  uint8_t* base_ = nullptr;
  size_t len_ = 0;
  int fd_ = -1;
};

}  // namespace nvdb
