! MODULE KFSP_CUSTOMPROP - how a COMPILED-IN propensity function reaches the device.
!
! The reference's own example drivers attach their propensities as a procedure pointer (MODEL%CUSTOMPROP,
! src/model/ModelModule.f90:163-199; examples/toggle.f90:55-69, repressilator.f90:50-69, transcr6d.f90:63-90): there
! is no expression to hand to kfsp_set_propensity_program.  What the device needs is a_k(x) for states the host never
! sees (the resident expansion step, DESIGN.md 10.6), so the function is PROBED once per solve and tabulated WITH THE
! USER'S OWN FUNCTION - every value the device ever uses is a value that function returned (or, for mass-action
! products, the same IEEE multiplications in the same order):
!   * which species does reaction k depend on?   a_k is evaluated at a set of base states with one population varied
!     at a time over 0 .. MAXNUMBERMOLECULES; bit patterns are compared
!   * none / one species  -> a table over every population count (tab_species / tab of kfsp_set_propensity_program)
!   * several species     -> first the product chains  c * x * y (* z)  in every order of the operands, c = a_k at
!                            populations 1: accepted if the chain reproduces a_k BIT FOR BIT on a lattice of
!                            populations (mass action; unbounded domain, no table)
!                         -> else, two species: a TWO-SPECIES TABLE (kfsp_set_propensity_tables2) over the populations
!                            seen so far plus a margin, enlarged (CUSTOM_GROW) when a device operation reports a
!                            population beyond it (-16) and the operation is repeated
!                         -> anything else: no plan, the model's propensities stay on the host
! The plan is SPECULATIVE - the probe cannot prove that a function ignores a species everywhere - so the caller
! (KRYLOVSOLVER) checks every propensity of the final lists against the function itself (CUSTOM_VERIFY) and repeats
! the solve with host propensities if a single bit differs.
MODULE KFSP_CUSTOMPROP
  USE MODELMODULE
  USE STATESPACE, ONLY: MAXNUMBERMOLECULES, FINITE_STATE_PROJECTION, CUSTOMPROP_IS_PURE
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: CUSTOM_PLAN, CUSTOM_PROBE, CUSTOM_ARRAYS, CUSTOM_GROW, CUSTOM_VERIFY

  INTEGER, PARAMETER :: MAXOPS = 4                       ! operands of a product chain (kPropMonoOps of the device)
  INTEGER(8), PARAMETER :: TABLE2_CAP = 67108864_8       ! doubles all two-species tables may hold together (512 MB)

  TYPE :: CUSTOM_PLAN
     LOGICAL :: OK = .FALSE.
     INTEGER :: NS = 0, NR = 0
     INTEGER, ALLOCATABLE :: KIND(:)          ! per reaction: 0 one-species table (or constant), 1 product chain, 2 two-species table
     INTEGER, ALLOCATABLE :: S1(:), S2(:)     ! species (1-based) of the table(s); S2 = 0 for KIND 0
     INTEGER, ALLOCATABLE :: N1(:), N2(:)     ! extents of a two-species table
     INTEGER, ALLOCATABLE :: NOPS(:), OPS(:, :)   ! chain: operand count, operands (0 = the constant, s = species s)
     DOUBLE PRECISION, ALLOCATABLE :: CONST(:)    ! chain: the constant
     INTEGER, ALLOCATABLE :: BASE(:)          ! the state the tables are made around (populations of the other species)
  END TYPE CUSTOM_PLAN

CONTAINS

  LOGICAL FUNCTION SAME_BITS(A, B)
    DOUBLE PRECISION, INTENT(IN) :: A, B
    SAME_BITS = TRANSFER(A, 1_8) == TRANSFER(B, 1_8)
  END FUNCTION SAME_BITS

  ! the next permutation of P(1:N) in lexicographic order; .FALSE. after the last
  LOGICAL FUNCTION NEXT_PERM(P, N)
    INTEGER, INTENT(INOUT) :: P(:)
    INTEGER, INTENT(IN) :: N
    INTEGER :: I, J, T
    NEXT_PERM = .FALSE.
    I = N - 1
    DO WHILE (I >= 1)
       IF (P(I) < P(I + 1)) EXIT
       I = I - 1
    ENDDO
    IF (I < 1) RETURN
    J = N
    DO WHILE (P(J) <= P(I))
       J = J - 1
    ENDDO
    T = P(I); P(I) = P(J); P(J) = T
    P(I + 1:N) = P(N:I + 1:-1)
    NEXT_PERM = .TRUE.
  END FUNCTION NEXT_PERM

  ! ((o1 * o2) * o3) ... as the device multiplies it out (prop_mono, csrc/kfsp_prop_dev.h)
  DOUBLE PRECISION FUNCTION CHAIN_VALUE(NOPS, OPS, C, X)
    INTEGER, INTENT(IN) :: NOPS, OPS(:), X(:)
    DOUBLE PRECISION, INTENT(IN) :: C
    INTEGER :: I
    DOUBLE PRECISION :: V, O
    V = 0.0D0
    DO I = 1, NOPS
       IF (OPS(I) == 0) THEN
          O = C
       ELSE
          O = DBLE(X(OPS(I)))
       ENDIF
       IF (I == 1) THEN
          V = O
       ELSE
          V = V * O
       ENDIF
    ENDDO
    CHAIN_VALUE = V
  END FUNCTION CHAIN_VALUE

  ! Probe MODEL%CUSTOMPROP around the seed states SEEDS(:, 1:NSEED) and make a plan (PLAN%OK = .FALSE.: none).
  SUBROUTINE CUSTOM_PROBE(MODEL, SEEDS, NSEED, PLAN)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, INTENT(IN) :: SEEDS(:, :), NSEED
    TYPE(CUSTOM_PLAN), INTENT(OUT) :: PLAN
    INTEGER, PARAMETER :: NV = 17, NG = 14, NBMAX = 10
    INTEGER, PARAMETER :: VALS(NV) = [0, 1, 2, 3, 4, 5, 7, 10, 16, 25, 40, 64, 100, 250, 1000, 4000, 10000]
    INTEGER, PARAMETER :: GRID(NG) = [0, 1, 2, 3, 5, 7, 11, 19, 37, 64, 101, 333, 1000, 9999]
    INTEGER :: NS, NR, NB, B, S, K, V, I, J, ND, DEPS(16), PERM(MAXOPS), NP, T, I1, I2, I3, TRY
    INTEGER, ALLOCATABLE :: BASES(:, :), X(:), RANGE(:)
    LOGICAL, ALLOCATABLE :: DEP(:, :)
    LOGICAL :: FOUND, GOOD
    DOUBLE PRECISION :: A0, A, C
    INTEGER(8) :: RNG
    INTEGER :: START
    CHARACTER(LEN=16) :: ENV
    NS = MODEL%NSPECIES
    NR = MODEL%NREACTIONS
    PLAN%OK = .FALSE.
    ! smallest extent a two-species table starts with (KFSP_CUSTOM_TABLE2_START; tests set it low to see the tables grow)
    START = 64
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_CUSTOM_TABLE2_START', ENV, I, J)
    IF (J == 0 .AND. I > 0) READ(ENV(1:I), *, IOSTAT=J) START
    START = MIN(MAX(START, 2), MAXNUMBERMOLECULES + 1)
    IF (.NOT. ASSOCIATED(MODEL%CUSTOMPROP) .OR. NS < 1 .OR. NS > 16 .OR. NR < 1 .OR. NR > 64 .OR. NSEED < 1) RETURN
    PLAN%NS = NS
    PLAN%NR = NR
    ALLOCATE(PLAN%KIND(NR), PLAN%S1(NR), PLAN%S2(NR), PLAN%N1(NR), PLAN%N2(NR), PLAN%NOPS(NR), PLAN%OPS(MAXOPS, NR), &
         PLAN%CONST(NR), PLAN%BASE(NS))
    PLAN%KIND = 0; PLAN%S1 = 1; PLAN%S2 = 0; PLAN%N1 = 0; PLAN%N2 = 0; PLAN%NOPS = 0; PLAN%OPS = 0; PLAN%CONST = 0.0D0
    ALLOCATE(BASES(NS, NBMAX), X(NS), RANGE(NS), DEP(NS, NR))
    ! base states: a few seeds, then lattice points from a fixed congruential sequence inside the seeds' bounding box
    ! stretched by a factor and a margin (every species away from 0 at least once: a factor x_s hides the others at 0)
    DO S = 1, NS
       RANGE(S) = MIN(MAXNUMBERMOLECULES, 3 * MAXVAL(SEEDS(S, 1:NSEED)) + 12)
    ENDDO
    NB = 0
    DO B = 1, MIN(NSEED, 3)
       NB = NB + 1
       BASES(:, NB) = SEEDS(1:NS, B)
    ENDDO
    RNG = 88172645463325252_8
    DO WHILE (NB < NBMAX)
       NB = NB + 1
       DO S = 1, NS
          RNG = IEOR(RNG, ISHFT(RNG, 13)); RNG = IEOR(RNG, ISHFT(RNG, -7)); RNG = IEOR(RNG, ISHFT(RNG, 17))
          BASES(S, NB) = 1 + INT(MODULO(RNG, INT(RANGE(S), 8)))
       ENDDO
    ENDDO
    PLAN%BASE = BASES(:, 1)
    ! dependencies, one population at a time
    DEP = .FALSE.
    DO K = 1, NR
       DO B = 1, NB
          X = BASES(:, B)
          A0 = MODEL%PROPENSITY(X, K)
          DO S = 1, NS
             IF (DEP(S, K)) CYCLE
             DO V = 1, NV
                X(S) = VALS(V)
                IF (.NOT. SAME_BITS(MODEL%PROPENSITY(X, K), A0)) THEN
                   DEP(S, K) = .TRUE.
                   EXIT
                ENDIF
             ENDDO
             X(S) = BASES(S, B)
          ENDDO
       ENDDO
    ENDDO
    DO K = 1, NR
       ND = 0
       DO S = 1, NS
          IF (DEP(S, K)) THEN
             ND = ND + 1
             DEPS(ND) = S
          ENDIF
       ENDDO
       IF (ND <= 1) THEN
          PLAN%KIND(K) = 0
          IF (ND == 1) PLAN%S1(K) = DEPS(1)
          CYCLE
       ENDIF
       ! a product chain?  operands: the constant (0) and the species; every order of them
       FOUND = .FALSE.
       IF (ND + 1 <= MAXOPS) THEN
          X = BASES(:, 1)
          X(DEPS(1:ND)) = 1
          C = MODEL%PROPENSITY(X, K)
          NP = ND + 1
          PERM(1) = 0
          PERM(2:NP) = DEPS(1:ND)
          ! (lexicographic permutations need a sorted start: 0 < every species index)
          DO
             GOOD = .TRUE.
             DO TRY = 1, 2
                X = BASES(:, TRY)
                DO I1 = 1, NG
                   DO I2 = 1, NG
                      DO I3 = 1, MERGE(NG, 1, ND == 3)
                         X(DEPS(1)) = GRID(I1)
                         X(DEPS(2)) = GRID(I2)
                         IF (ND == 3) X(DEPS(3)) = GRID(I3)
                         IF (.NOT. SAME_BITS(CHAIN_VALUE(NP, PERM, C, X), MODEL%PROPENSITY(X, K))) THEN
                            GOOD = .FALSE.
                            EXIT
                         ENDIF
                      ENDDO
                      IF (.NOT. GOOD) EXIT
                   ENDDO
                   IF (.NOT. GOOD) EXIT
                ENDDO
                IF (.NOT. GOOD) EXIT
             ENDDO
             IF (GOOD) THEN
                FOUND = .TRUE.
                EXIT
             ENDIF
             IF (.NOT. NEXT_PERM(PERM, NP)) EXIT
          ENDDO
          IF (FOUND) THEN
             PLAN%KIND(K) = 1
             PLAN%NOPS(K) = NP
             PLAN%OPS(1:NP, K) = PERM(1:NP)
             PLAN%CONST(K) = C
             CYCLE
          ENDIF
       ENDIF
       IF (ND /= 2) RETURN                                  ! three and more species, not a product: the host keeps this model
       PLAN%KIND(K) = 2
       PLAN%S1(K) = DEPS(1)
       PLAN%S2(K) = DEPS(2)
       PLAN%N1(K) = MIN(MAXNUMBERMOLECULES + 1, MAX(START, 4 * MAXVAL(SEEDS(DEPS(1), 1:NSEED)) + START / 2))
       PLAN%N2(K) = MIN(MAXNUMBERMOLECULES + 1, MAX(START, 4 * MAXVAL(SEEDS(DEPS(2), 1:NSEED)) + START / 2))
    ENDDO
    ! the plan's value against the function at lattice points where ALL populations vary at once
    DO T = 1, 400
       DO S = 1, NS
          RNG = IEOR(RNG, ISHFT(RNG, 13)); RNG = IEOR(RNG, ISHFT(RNG, -7)); RNG = IEOR(RNG, ISHFT(RNG, 17))
          X(S) = INT(MODULO(RNG, INT(RANGE(S) + 1, 8)))
       ENDDO
       DO K = 1, NR
          A = MODEL%PROPENSITY(X, K)
          IF (.NOT. SAME_BITS(A, PLAN_VALUE(MODEL, PLAN, K, X))) RETURN
       ENDDO
    ENDDO
    PLAN%OK = .TRUE.
  END SUBROUTINE CUSTOM_PROBE

  ! a_k(x) as the device will form it from the plan (tables are made of calls with the other populations at PLAN%BASE)
  DOUBLE PRECISION FUNCTION PLAN_VALUE(MODEL, PLAN, K, X)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    TYPE(CUSTOM_PLAN), INTENT(IN) :: PLAN
    INTEGER, INTENT(IN) :: K, X(:)
    INTEGER :: Y(PLAN%NS)
    SELECT CASE (PLAN%KIND(K))
    CASE (1)
       PLAN_VALUE = CHAIN_VALUE(PLAN%NOPS(K), PLAN%OPS(:, K), PLAN%CONST(K), X)
    CASE (2)
       Y = PLAN%BASE
       Y(PLAN%S1(K)) = X(PLAN%S1(K))
       Y(PLAN%S2(K)) = X(PLAN%S2(K))
       PLAN_VALUE = MODEL%PROPENSITY(Y, K)
    CASE DEFAULT
       Y = PLAN%BASE
       Y(PLAN%S1(K)) = X(PLAN%S1(K))
       PLAN_VALUE = MODEL%PROPENSITY(Y, K)
    END SELECT
  END FUNCTION PLAN_VALUE

  ! The arrays of kfsp_set_propensity_program / kfsp_set_propensity_tables2 (include/kfsp.h) for the plan: chains as
  ! postfix code  o1 o2 MUL o3 MUL ...  (the device recognises them and multiplies them out), one-species tables over
  ! 0 .. MAXNUMBERMOLECULES, two-species tables over their current extents.  All made with MODEL%PROPENSITY.
  SUBROUTINE CUSTOM_ARRAYS(MODEL, PLAN, CODE_OFF, CODE, IMM_OFF, IMM, TAB_SPECIES, TAB_LEN, TAB, T2S1, T2S2, T2N1, T2N2, T2OFF, T2LEN, &
       TAB2)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    TYPE(CUSTOM_PLAN), INTENT(IN) :: PLAN
    INTEGER, ALLOCATABLE, INTENT(OUT) :: CODE_OFF(:), CODE(:), IMM_OFF(:), TAB_SPECIES(:), T2S1(:), T2S2(:), T2N1(:), T2N2(:)
    INTEGER(8), ALLOCATABLE, INTENT(OUT) :: T2OFF(:)
    INTEGER(8), INTENT(OUT) :: T2LEN
    INTEGER, INTENT(OUT) :: TAB_LEN
    DOUBLE PRECISION, ALLOCATABLE, INTENT(OUT) :: IMM(:), TAB(:, :), TAB2(:)
    INTEGER :: K, I, NC, NI, V, V2, NR
    INTEGER :: X(PLAN%NS)
    INTEGER(8) :: P
    NR = PLAN%NR
    ALLOCATE(CODE_OFF(NR + 1), IMM_OFF(NR + 1), TAB_SPECIES(NR), T2S1(NR), T2S2(NR), T2N1(NR), T2N2(NR), T2OFF(NR))
    NC = 0
    NI = 0
    DO K = 1, NR
       CODE_OFF(K) = NC
       IMM_OFF(K) = NI
       IF (PLAN%KIND(K) == 1) THEN
          NC = NC + 2 * PLAN%NOPS(K) - 1
          NI = NI + COUNT(PLAN%OPS(1:PLAN%NOPS(K), K) == 0)
       ENDIF
    ENDDO
    CODE_OFF(NR + 1) = NC
    IMM_OFF(NR + 1) = NI
    ALLOCATE(CODE(MAX(NC, 1)), IMM(MAX(NI, 1)))
    CODE = 0
    IMM = 0.0D0
    TAB_LEN = MAXNUMBERMOLECULES + 1
    ALLOCATE(TAB(TAB_LEN, NR))
    TAB = 0.0D0
    T2LEN = 0
    DO K = 1, NR
       TAB_SPECIES(K) = -1
       T2S1(K) = -1; T2S2(K) = -1; T2N1(K) = 0; T2N2(K) = 0; T2OFF(K) = 0
       SELECT CASE (PLAN%KIND(K))
       CASE (1)
          NC = CODE_OFF(K)
          NI = IMM_OFF(K)
          DO I = 1, PLAN%NOPS(K)
             NC = NC + 1
             IF (PLAN%OPS(I, K) == 0) THEN
                CODE(NC) = 1                                  ! IMM
                NI = NI + 1
                IMM(NI) = PLAN%CONST(K)
             ELSE
                CODE(NC) = 100 + PLAN%OPS(I, K)               ! species variable
             ENDIF
             IF (I >= 2) THEN
                NC = NC + 1
                CODE(NC) = 5                                  ! MUL
             ENDIF
          ENDDO
       CASE (2)
          T2S1(K) = PLAN%S1(K) - 1
          T2S2(K) = PLAN%S2(K) - 1
          T2N1(K) = PLAN%N1(K)
          T2N2(K) = PLAN%N2(K)
          T2OFF(K) = T2LEN
          T2LEN = T2LEN + INT(PLAN%N1(K), 8) * INT(PLAN%N2(K), 8)
       CASE DEFAULT
          TAB_SPECIES(K) = PLAN%S1(K) - 1
          X = PLAN%BASE
          DO V = 0, TAB_LEN - 1
             X(PLAN%S1(K)) = V
             TAB(V + 1, K) = MODEL%PROPENSITY(X, K)
          ENDDO
       END SELECT
    ENDDO
    ALLOCATE(TAB2(MAX(T2LEN, 1_8)))
    TAB2 = 0.0D0
    DO K = 1, NR
       IF (PLAN%KIND(K) /= 2) CYCLE
       X = PLAN%BASE
       P = T2OFF(K)
       DO V2 = 0, PLAN%N2(K) - 1
          X(PLAN%S2(K)) = V2
          DO V = 0, PLAN%N1(K) - 1
             X(PLAN%S1(K)) = V
             P = P + 1
             TAB2(P) = MODEL%PROPENSITY(X, K)
          ENDDO
       ENDDO
    ENDDO
  END SUBROUTINE CUSTOM_ARRAYS

  ! MISSED(s) = largest population of species s a device operation found beyond a two-species table
  ! (kfsp_propensity_overflow): the tables of that species at least double.  .FALSE.: they would exceed TABLE2_CAP.
  LOGICAL FUNCTION CUSTOM_GROW(PLAN, MISSED)
    TYPE(CUSTOM_PLAN), INTENT(INOUT) :: PLAN
    INTEGER, INTENT(IN) :: MISSED(:)
    INTEGER :: K
    INTEGER(8) :: TOTAL
    LOGICAL :: GREW
    GREW = .FALSE.
    DO K = 1, PLAN%NR
       IF (PLAN%KIND(K) /= 2) CYCLE
       IF (MISSED(PLAN%S1(K)) >= PLAN%N1(K)) THEN
          PLAN%N1(K) = MIN(MAXNUMBERMOLECULES + 1, MAX(2 * PLAN%N1(K), MISSED(PLAN%S1(K)) + 17))
          GREW = .TRUE.
       ENDIF
       IF (MISSED(PLAN%S2(K)) >= PLAN%N2(K)) THEN
          PLAN%N2(K) = MIN(MAXNUMBERMOLECULES + 1, MAX(2 * PLAN%N2(K), MISSED(PLAN%S2(K)) + 17))
          GREW = .TRUE.
       ENDIF
    ENDDO
    TOTAL = 0
    DO K = 1, PLAN%NR
       IF (PLAN%KIND(K) == 2) TOTAL = TOTAL + INT(PLAN%N1(K), 8) * INT(PLAN%N2(K), 8)
    ENDDO
    CUSTOM_GROW = GREW .AND. TOTAL <= TABLE2_CAP
  END FUNCTION CUSTOM_GROW

  ! every propensity of the lists against the function itself: the number of entries whose bits differ
  INTEGER FUNCTION CUSTOM_VERIFY(MODEL, FSP) RESULT(NBAD)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    TYPE(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER :: I, K, NS, NR
    DOUBLE PRECISION :: D
    LOGICAL :: PAR
    NS = MODEL%NSPECIES
    NR = MODEL%NREACTIONS
    NBAD = 0
    PAR = CUSTOMPROP_IS_PURE() .AND. FSP%SIZE > 4096
    !$OMP PARALLEL DO SCHEDULE(STATIC) IF(PAR) PRIVATE(K, D) REDUCTION(+:NBAD)
    DO I = 1, FSP%SIZE
       D = 0.0D0
       DO K = 1, NR
          IF (.NOT. SAME_BITS(MODEL%PROPENSITY(FSP%STATE(1:NS, I), K), FSP%MATRIX%OFFDIAG(K, I))) NBAD = NBAD + 1
          D = D + FSP%MATRIX%OFFDIAG(K, I)
       ENDDO
       IF (.NOT. SAME_BITS(D, FSP%MATRIX%DIAG(I))) NBAD = NBAD + 1
    ENDDO
    !$OMP END PARALLEL DO
  END FUNCTION CUSTOM_VERIFY

END MODULE KFSP_CUSTOMPROP
