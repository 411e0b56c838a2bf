/* reloc_spec.h -- numeric constants of the relocalization hot path.
 *
 * Constants only (no arithmetic): the CPU oracle (oracle/) and the HIP product library
 * (nclt-slam-project_amd/csrc/) each implement the stages independently and are held
 * bit-exact to each other by tests/.  Every constant cites the reference call site that
 * fixes it (paths relative to the reference repository root) or, where the arithmetic
 * lives inside OpenCV (absent from the reference tree and from this container), the
 * SURVEY.md Appendix A paragraph that restates OpenCV's published algorithm.
 */
#ifndef RELOC_SPEC_H
#define RELOC_SPEC_H

/* ORB_create(nfeatures=500): every other parameter left at OpenCV's default
 * (simulation/isaac/scripts/common/visual_landmark_matcher.py:207,
 *  simulation/isaac/scripts/common/visual_landmark_recorder.py:159). */
#define RELOC_ORB_NLEVELS        8
#define RELOC_ORB_SCALE_FACTOR   1.2          /* level scale = (float)pow(1.2, level)          */
#define RELOC_ORB_EDGE           31           /* edgeThreshold: keep 31 <= x < w-31             */
#define RELOC_ORB_PATCH          31           /* KeyPoint.size = 31 * scale                     */
#define RELOC_ORB_HALF_PATCH     15           /* intensity-centroid disc radius                 */
#define RELOC_FAST_THRESHOLD     20           /* fastThreshold                                  */
#define RELOC_FAST_ARC           9            /* FAST-9/16                                      */
#define RELOC_HARRIS_BLOCK       7            /* HARRIS_BLOCK_SIZE                              */
#define RELOC_HARRIS_K           0.04f
/* Capacity of the per-level "best 2*quota by FAST score, ties kept" set.  If more pixels
 * than this reach the cut score the cut is raised one score at a time until the set fits
 * (decided from the score histogram alone, so the rule is order-independent). */
#define RELOC_ORB_STAGE1_CAP     4096

/* BGR->gray 8-bit fixed point.  Two published OpenCV conventions, chosen by reloc_params.gray_coeff_bits:
 *   15 (DEFAULT since round 4; OpenCV 4.x 8-bit path, `gray_shift`): Y = (B*3735 + G*19235 + R*9798 + 16384) >> 15
 *   14 (OpenCV <= 3.x, `yuv_shift`):                  Y = (B*1868 + G*9617 + R*4899 + 8192)  >> 14
 * The reference can only run on OpenCV >= 4.8 (datasets/nclt/requirements.txt:3; ROS 2 Jazzy / NumPy-2-era wheel: >= 4.10), so the
 * call it makes at M:305 / R:240 computes the 15-bit form.  The two differ by +-1 on some pixels.  Neither can be checked against
 * OpenCV offline (parity unpinned). */
#define RELOC_GRAY_CB            1868
#define RELOC_GRAY_CG            9617
#define RELOC_GRAY_CR            4899
#define RELOC_GRAY_SHIFT         14
#define RELOC_GRAY15_CB          3735
#define RELOC_GRAY15_CG          19235
#define RELOC_GRAY15_CR          9798
#define RELOC_GRAY15_SHIFT       15
#define RELOC_GRAY_DEFAULT_BITS  RELOC_GRAY15_SHIFT
/* order argument of the gray stage: bit 0 = channel order (RELOC_ORDER_RGB), bit 1 = the 15-bit coefficient set */
#define RELOC_GRAY_FLAG_15BIT    2

/* 7x7 sigma=2 Gaussian in 8 fractional bits, sum == 256.  Horizontal pass in
 * 8.8 fixed point, vertical pass in 16.16, result = (v + 32768) >> 16, BORDER_REFLECT_101. */
#define RELOC_BLUR_K0            18
#define RELOC_BLUR_K1            33
#define RELOC_BLUR_K2            49
#define RELOC_BLUR_K3            56

/* INTER_LINEAR_EXACT restatement: 8 fractional bits per axis coefficient. */
#define RELOC_RESIZE_COEF_BITS   8

/* fastAtan2 polynomial, degrees; evaluated in float without fused multiply-add. */
#define RELOC_ATAN2_P1           57.283627f   /*  0.9997878412794807 * 180/pi */
#define RELOC_ATAN2_P3          (-18.667446f) /* -0.3258083974640975 * 180/pi */
#define RELOC_ATAN2_P5           8.9140005f   /*  0.1555786518463281 * 180/pi */
#define RELOC_ATAN2_P7          (-2.5397246f) /* -0.04432655554792128 * 180/pi */
#define RELOC_ATAN2_EPS          2.220446049250313e-16f  /* (float)DBL_EPSILON */
#define RELOC_DEG2RAD_F          0.017453292519943295f   /* (float)(pi/180)     */

/* Matcher gates (visual_landmark_matcher.py:56-76). */
#define RELOC_CANDIDATE_RADIUS_M 8.0
#define RELOC_MAX_CANDIDATES     5
#define RELOC_HEADING_TOL_DEG    90.0
#define RELOC_MIN_MATCHES        10
#define RELOC_MIN_INLIERS        10
#define RELOC_REPROJ_MAX_PX      2.0
#define RELOC_RANSAC_REPROJ_PX   3.0
#define RELOC_RANSAC_ITERATIONS  200
#define RELOC_RANSAC_CONFIDENCE  0.99         /* cv2.solvePnPRansac default */
#define RELOC_CONSISTENCY_M      5.0
/* Global-relocalisation variant (experiments/63_global_reloc/scripts/visual_landmark_matcher.py) */
#define RELOC_GLOBAL_MAX_CANDIDATES 25
#define RELOC_GLOBAL_MIN_INLIERS    18        /* :85 */
#define RELOC_GLOBAL_REPROJ_MAX_PX  1.5       /* :86 */
/* Accumulation (visual_landmark_matcher.py:85-89, :461). */
#define RELOC_ACCUM_MIN_DIST_M   5.0
#define RELOC_ACCUM_MIN_KPTS     30
#define RELOC_ACCUM_DEPTH_MIN_M  0.5
#define RELOC_ACCUM_DEPTH_MAX_M  15.0

/* Pinhole intrinsics (visual_landmark_matcher.py:49-52, visual_landmark_recorder.py:55-57). */
#define RELOC_FX 320.0
#define RELOC_FY 320.0
#define RELOC_CX 320.0
#define RELOC_CY 240.0

/* Recorder gates (visual_landmark_recorder.py:59-71, :270). */
#define RELOC_DEPTH_MIN_M        0.5f
#define RELOC_DEPTH_MAX_M        15.0f
#define RELOC_DEPTH_VAR_MAX_M    0.30f
#define RELOC_GROUND_Y_THRESHOLD 180
#define RELOC_MIN_RECORD_KPTS    30

/* Hypothesis sampler shared by oracle and product so both score the same hypothesis list
 * (SURVEY.md A.8 hazard 5): splitmix64 counter stream, four distinct indices per hypothesis. */
#define RELOC_RNG_GOLDEN         0x9E3779B97F4A7C15ull
#define RELOC_RNG_MUL1           0xBF58476D1CE4E5B9ull
#define RELOC_RNG_MUL2           0x94D049BB133111EBull
#define RELOC_PNP_SAMPLE         4            /* 3 for P3P + 1 to pick among its <=4 roots */
#define RELOC_LM_MAX_TRIALS      30
#define RELOC_LM_LAMBDA0         1e-3
/* Levenberg-Marquardt stops after the step that is smaller than STEP_EPS (rad / m) or changes the cost by less than
 * COST_EPS * cost.  The iteration converges quadratically from a RANSAC pose (steps 1e-2, 5e-5, 5e-8, 3e-11 on typical
 * problems), so a step below 1e-4 leaves an error of ~1e-8 -- four orders below the 1e-4 m / 1e-4 rad tolerance of the
 * north star; rounds 1-2 iterated to 1e-10 / 1e-13, two more trials (~7 us of a single-wave kernel) for digits nobody reads. */
#define RELOC_LM_STEP_EPS        1e-4
#define RELOC_LM_COST_EPS        1e-8         /* stop when |cost change| <= this * cost */
/* Resolvent cubic of the P3P quartic: Halley iteration from a Fujiwara-type upper bound, inside the bracket [0, hi] with a
 * bisection fallback, at most this many steps (rounds 1-2: Newton from the coefficient bound until x stopped moving, up to
 * 80 steps -- mean 32, 95th percentile 68, and a wave waits for its slowest lane).  The quartic's roots are polished on the
 * quartic itself afterwards, so the resolvent root needs no more. */
#define RELOC_P3P_CUBIC_ITERS    16

#endif /* RELOC_SPEC_H */
