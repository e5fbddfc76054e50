#include "nvdb/mmap_file.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <stdexcept>

namespace nvdb {

namespace {
[[noreturn]] void fail_errno(const char* call) { throw std::runtime_error(std::string(call) + ": " + std::strerror(errno)); }
}  // namespace

void MmapFile::open_readonly(const std::string& path) {
  reset();
  const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0) fail_errno("open");
  struct stat st;
  if (::fstat(fd, &st) != 0) { ::close(fd); fail_errno("fstat"); }
  if (st.st_size <= 0) { ::close(fd); throw std::runtime_error("File size is zero"); }
  void* p = ::mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
  if (p == MAP_FAILED) { ::close(fd); fail_errno("mmap"); }
  base_ = static_cast<uint8_t*>(p);
  len_ = static_cast<size_t>(st.st_size);
  fd_ = fd;
}

void MmapFile::reset() {
  if (base_) ::munmap(base_, len_);
  if (fd_ >= 0) ::close(fd_);
  base_ = nullptr; len_ = 0; fd_ = -1;
}

}  // namespace nvdb
