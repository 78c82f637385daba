/* tests/stubs/mex.h -- a minimal, hand-written stand-in for MATLAB's mex.h / matrix.h, for ONE purpose: letting
 * `g++ -fsyntax-only` parse and type-check the gateways under mex/ in an image that has no MATLAB (tests/test_binding_syntax.py).
 * It declares only the mx* / mex* functions, types and enumerators the gateways use, with the signatures of the
 * published MATLAB C Matrix / MEX API.  Nothing here is ever linked or run, and a gateway that compiles against it is
 * syntactically sound C++ that calls these functions with plausible types -- not evidence of parity or of a working
 * MEX build (INTEGRATION.md has the real build line). */
#ifndef VO_TEST_STUB_MEX_H
#define VO_TEST_STUB_MEX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;

typedef enum {
    mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS,
    mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS,
    mxINT64_CLASS, mxUINT64_CLASS, mxFUNCTION_CLASS
} mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;

/* the entry point every gateway defines */
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

/* MEX library */
void mexErrMsgIdAndTxt(const char* identifier, const char* err_msg, ...);
void mexLock(void);
int mexAtExit(void (*exit_fcn)(void));

/* queries */
double* mxGetPr(const mxArray* pa);
void* mxGetData(const mxArray* pa);
double mxGetScalar(const mxArray* pa);
size_t mxGetM(const mxArray* pa);
size_t mxGetN(const mxArray* pa);
size_t mxGetNumberOfElements(const mxArray* pa);
mwSize mxGetNumberOfDimensions(const mxArray* pa);
const mwSize* mxGetDimensions(const mxArray* pa);
bool mxIsDouble(const mxArray* pa);
bool mxIsSingle(const mxArray* pa);
bool mxIsUint8(const mxArray* pa);
bool mxIsUint64(const mxArray* pa);
mxArray* mxGetField(const mxArray* pa, mwIndex i, const char* fieldname);
mxArray* mxGetCell(const mxArray* pa, mwIndex i);
int mxGetString(const mxArray* pa, char* buf, mwSize buflen);
char* mxArrayToString(const mxArray* pa);

/* creation / mutation */
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double value);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag);
mxArray* mxCreateLogicalMatrix(mwSize m, mwSize n);
mxArray* mxCreateLogicalScalar(bool value);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames);
mxArray* mxCreateCellMatrix(mwSize m, mwSize n);
void mxSetField(mxArray* pa, mwIndex i, const char* fieldname, mxArray* value);
void mxSetCell(mxArray* pa, mwIndex i, mxArray* value);
void mxDestroyArray(mxArray* pa);
void* mxMalloc(size_t n);
void mxFree(void* ptr);

#ifdef __cplusplus
}
#endif
#endif
