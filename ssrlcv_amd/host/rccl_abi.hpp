// ssrlcv_amd/host/rccl_abi.hpp -- the handful of RCCL / HIP runtime entry points Distributed.hpp calls, declared by hand.
// RCCL's own header pulls in HIP's vector types (float2, float4, uint2 ... as templates), which collide with the POD
// types of the same names the mirror defines for layout compatibility with the reference (cuda_vec_types.hpp).  The
// declarations below restate /opt/rocm/include/rccl/rccl.h (ROCm 7.2: lines 40-56, 187, 220, 260, 339, 448-470, 591-612,
// 923-933) and hip_runtime_api.h; they are C functions with a stable ABI (the NCCL 2.x one).
#pragma once
#include <stddef.h>

extern "C" {
typedef struct ihipStream_t* hipStream_t;
int hipSetDevice(int deviceId);                   // hipError_t is an int-sized enum, 0 = hipSuccess
int hipStreamSynchronize(hipStream_t stream);
const char* hipGetErrorString(int hipError);

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;  // NCCL_UNIQUE_ID_BYTES
typedef enum { ncclSuccess = 0 } ncclResult_t;        // every other value is a failure; ncclGetErrorString names it
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclChar = 0, ncclUint32 = 3, ncclFloat32 = 7 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char* ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
}
