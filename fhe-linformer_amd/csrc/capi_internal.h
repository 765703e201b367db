// Shared plumbing for the extern "C" translation units.
#pragma once
#include <exception>
#include <string>
#include "context.h"
#include "evaluator.h"
#include "client.h"
#include "composite.h"
#include "bootstrap.h"

// opaque handle behind include/fhelin.h's `fhelin_ctx`
struct fhelin_ctx {
    fhelin::Context ctx;
    fhelin::Evaluator ev;
    fhelin::Client cl;
    fhelin::Composite comp;
    fhelin::Bootstrapper boot;
    explicit fhelin_ctx(const fhelin::Params& p) : ctx(p), ev(ctx), cl(ev, p.seed), comp(ev, cl), boot(ev, cl) {}
};
struct fhelin_ct {
    fhelin::CtPtr p;
};
struct fhelin_pt {
    fhelin::PtPtr p;
};

namespace fhelin {
int capi_fail(int code, const std::string& msg);
}

#define FHELIN_TRY try {
#define FHELIN_CATCH                                                          \
    return FHELIN_OK;                                                         \
    }                                                                         \
    catch (const fhelin::Error& e) { return fhelin::capi_fail(e.code, e.what()); }          \
    catch (const std::bad_alloc&) { return fhelin::capi_fail(FHELIN_ERR_INTERNAL, "host out of memory"); } \
    catch (const std::exception& e) { return fhelin::capi_fail(FHELIN_ERR_INTERNAL, e.what()); }
