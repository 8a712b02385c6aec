// The slice of RCCL's C ABI that comm.hip binds at run time (dlopen): result codes, the 128-byte unique
// id, data-type / reduction enumerators and the function-pointer shapes of the ten entry points used.
// Declared here so that libepa_dev.so builds on a ROCm install without the RCCL development headers
// (RCCL is optional at run time as well: a process that never creates a communicator never loads it).
// Values as published in rccl.h / nccl.h (stable since NCCL 2.x).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

extern "C" {
typedef struct ncclComm* ncclComm_t;
#define EPA_NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[EPA_NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum {
  ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
  ncclInvalidArgument = 4, ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7
} ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
typedef enum {
  ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4,
  ncclUint64 = 5, ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8
} ncclDataType_t;
}

struct epa_rccl_api {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
