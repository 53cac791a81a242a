/*
 * timing_info.h -- drop-in for the reference's timing_info.h: same TimingInfo type
 * (timing_info.h:6-12), same table layout when printed (timing_info.h:40-47).
 *
 * Two generations of call sites exist in the reference tree and both are accepted
 * (selected by argument count):
 *   current (timing_info.h:14,34,40,69)            legacy (test_mg_3d_dirichlet.c:27,71,101;
 *                                                          mg_dirichlet_analytic.c:96,1007,1036)
 *   allocTimingInfo(&t, stageNames, numStages)     allocTimingInfo(&tInfo, numLevels)
 *   printTimingInfo(t)                             printTimingInfo(tInfo, numLevels)
 *   deAllocTimingInfo(&t)                          deAllocTimingInfo(&tInfo, numLevels)
 *   resetTimingInfo(t)                             resetTimingInfo(tInfo, numLevels)
 * In the legacy form tInfo is the per-level array (TimingInfo **), each level carrying the
 * seven V-cycle stages of mg_3d.h:136-137.  The numbers come from hipEvent pairs recorded
 * by libmg3d around each stage (mg3d_timing_get), not from omp_get_wtime.
 */
#ifndef TIMING_INFO_H
#define TIMING_INFO_H

#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h> /* clock()/CLOCKS_PER_SEC for the older drivers (test_lu.c:37-41) */

#include "mg3d.h"

/* Per-level accessors used by mg_3d.h (same spelling for both struct generations below). */
#ifndef MG3D_LEGACY_TIMINGINFO

typedef struct __time_t {
    int numStages;
    char **stageNames;
    int *numCalls;
    double *timeTaken;
} TimingInfo;

#define MG3D_TI_NSTAGES_(ti, l) ((ti)[l]->numStages)
#define MG3D_TI_CALLS_(ti, l, s) ((ti)[l]->numCalls[s])
#define MG3D_TI_TIME_(ti, l, s) ((ti)[l]->timeTaken[s])

static inline void mg3d_timing_alloc_one_(TimingInfo **t, const char *const *stageNames, int numStages)
{
    *t = (TimingInfo *)malloc(sizeof(TimingInfo));
    assert(*t);
    (*t)->numStages = numStages;
    (*t)->stageNames = (char **)malloc(sizeof(char *) * (size_t)numStages);
    for (int i = 0; i < numStages; i++)
        (*t)->stageNames[i] = strdup(stageNames[i]);
    (*t)->numCalls = (int *)calloc((size_t)numStages, sizeof(int));
    (*t)->timeTaken = (double *)calloc((size_t)numStages, sizeof(double));
}

static inline void mg3d_timing_alloc_levels_(TimingInfo ***tInfo, int numLevels)
{
    const char *names[MG3D_NUM_STAGES];
    for (int s = 0; s < MG3D_NUM_STAGES; s++)
        names[s] = mg3d_stage_name(s);
    *tInfo = (TimingInfo **)malloc(sizeof(TimingInfo *) * (size_t)numLevels);
    assert(*tInfo);
    for (int l = 0; l < numLevels; l++)
        mg3d_timing_alloc_one_(&(*tInfo)[l], names, MG3D_NUM_STAGES);
}

static inline void mg3d_timing_reset_one_(TimingInfo *t)
{
    memset(t->numCalls, 0, sizeof(int) * (size_t)t->numStages);
    memset(t->timeTaken, 0, sizeof(double) * (size_t)t->numStages);
}

static inline void mg3d_timing_reset_levels_(TimingInfo **tInfo, int numLevels)
{
    for (int l = 0; l < numLevels; l++)
        mg3d_timing_reset_one_(tInfo[l]);
}

static inline void mg3d_timing_print_one_(TimingInfo *t)
{
    printf("%20s %20s %20s\n", "", "numCalls", "timeTaken");
    for (int i = 0; i < t->numStages; i++)
        printf("%20.20s %20d %20lf\n", t->stageNames[i], t->numCalls[i], t->timeTaken[i]);
}

static inline void mg3d_timing_print_levels_(TimingInfo **tInfo, int numLevels)
{
    for (int l = 0; l < numLevels; l++) {
        printf("LEVEL %d\n", l);
        mg3d_timing_print_one_(tInfo[l]);
    }
}

static inline void mg3d_timing_free_one_(TimingInfo **t)
{
    for (int i = 0; i < (*t)->numStages; i++)
        free((*t)->stageNames[i]);
    free((*t)->stageNames);
    free((*t)->numCalls);
    free((*t)->timeTaken);
    free(*t);
    *t = NULL;
}

static inline void mg3d_timing_free_levels_(TimingInfo ***tInfo, int numLevels)
{
    for (int l = 0; l < numLevels; l++)
        mg3d_timing_free_one_(&(*tInfo)[l]);
    free(*tInfo);
    *tInfo = NULL;
}

#else /* MG3D_LEGACY_TIMINGINFO */

/* The struct generation the older drivers of the reference tree were written against:
 * one {numCalls, timeTaken} record per (level, stage), tInfo[level][stage]
 * (mg_dirichlet_analytic.c:712-766), or a single record used as a stopwatch
 * (test_rb_gs_3d.c:22-24,72-73,117; test_gs_3d.c:22-24).  Build those drivers with
 * -DMG3D_LEGACY_TIMINGINFO; only the per-level (2-argument) helper forms exist here. */
typedef struct __time_t {
    int numCalls;
    double timeTaken;
} TimingInfo;

#define MG3D_TI_NSTAGES_(ti, l) (MG3D_NUM_STAGES)
#define MG3D_TI_CALLS_(ti, l, s) ((ti)[l][s].numCalls)
#define MG3D_TI_TIME_(ti, l, s) ((ti)[l][s].timeTaken)

static inline void mg3d_timing_alloc_levels_(TimingInfo ***tInfo, int numLevels)
{
    *tInfo = (TimingInfo **)malloc(sizeof(TimingInfo *) * (size_t)numLevels);
    assert(*tInfo);
    for (int l = 0; l < numLevels; l++) {
        (*tInfo)[l] = (TimingInfo *)calloc(MG3D_NUM_STAGES, sizeof(TimingInfo));
        assert((*tInfo)[l]);
    }
}

static inline void mg3d_timing_reset_levels_(TimingInfo **tInfo, int numLevels)
{
    for (int l = 0; l < numLevels; l++)
        memset(tInfo[l], 0, sizeof(TimingInfo) * MG3D_NUM_STAGES);
}

static inline void mg3d_timing_print_levels_(TimingInfo **tInfo, int numLevels)
{
    for (int l = 0; l < numLevels; l++) {
        printf("LEVEL %d\n", l);
        printf("%20s %20s %20s\n", "", "numCalls", "timeTaken");
        for (int s = 0; s < MG3D_NUM_STAGES; s++)
            printf("%20.20s %20d %20lf\n", mg3d_stage_name(s), tInfo[l][s].numCalls, tInfo[l][s].timeTaken);
    }
}

static inline void mg3d_timing_free_levels_(TimingInfo ***tInfo, int numLevels)
{
    for (int l = 0; l < numLevels; l++)
        free((*tInfo)[l]);
    free(*tInfo);
    *tInfo = NULL;
}

#endif /* MG3D_LEGACY_TIMINGINFO */

/* argument-count dispatch (C has no overloading; function-like macros do not capture
 * identifiers, so locals named like the reference's globals stay untouched) */
#ifndef MG3D_LEGACY_TIMINGINFO
#define MG3D_PICK3_(a, b, c, name, ...) name
#define MG3D_PICK2_(a, b, name, ...) name
#define allocTimingInfo(...) \
    MG3D_PICK3_(__VA_ARGS__, mg3d_timing_alloc3_, mg3d_timing_alloc_levels_, )(__VA_ARGS__)
#define mg3d_timing_alloc3_(t, names, n) mg3d_timing_alloc_one_((t), (const char *const *)(names), (n))
#define printTimingInfo(...) MG3D_PICK2_(__VA_ARGS__, mg3d_timing_print_levels_, mg3d_timing_print_one_, )(__VA_ARGS__)
#define resetTimingInfo(...) MG3D_PICK2_(__VA_ARGS__, mg3d_timing_reset_levels_, mg3d_timing_reset_one_, )(__VA_ARGS__)
#define deAllocTimingInfo(...) MG3D_PICK2_(__VA_ARGS__, mg3d_timing_free_levels_, mg3d_timing_free_one_, )(__VA_ARGS__)
#else
#define allocTimingInfo(t, n) mg3d_timing_alloc_levels_((t), (n))
#define printTimingInfo(t, n) mg3d_timing_print_levels_((t), (n))
#define resetTimingInfo(t, n) mg3d_timing_reset_levels_((t), (n))
#define deAllocTimingInfo(t, n) mg3d_timing_free_levels_((t), (n))
#endif

#endif
