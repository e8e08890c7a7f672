! Propensity expressions of models/*.input: string -> postfix code -> value.
!
! Host-side, runs once per state at assembly time (not part of the HIP path).
! Behaviour follows what the reference's expression type does for the model
! files (src/parser/FortranParser.f90:172-302 evaluation, :627-723 operator
! splitting), restated from its observable rules:
!   * blanks are ignored, '**' means '^', names are case-sensitive and may
!     contain '.' and digits (DNA.2D), function names are not case-sensitive
!   * a sub-expression is split at the RIGHTMOST top-level binary operator of
!     the lowest class present, the classes being tried in the order
!     + , - , * , / , ^   (so a*b/c is a*(b/c) and a+b-c is a+(b-c))
!   * a leading '-' in front of a product/quotient/power negates the whole of it
!   * x/0 evaluates the whole expression to 0; log/log10 of x<=0, sqrt of x<0,
!     asin/acos outside [-1,1] likewise
!   * numbers accept d/D/e/E exponents
MODULE KFSP_EXPR
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: EXPRESSION, EXPR_COMPILE, EXPR_EVAL

  INTEGER, PARAMETER :: OP_IMM = 1, OP_NEG = 2, OP_ADD = 3, OP_SUB = 4, OP_MUL = 5, OP_DIV = 6, OP_POW = 7
  INTEGER, PARAMETER :: OP_FUN = 10      ! OP_FUN + k, k = 1..14
  INTEGER, PARAMETER :: OP_VAR = 100     ! OP_VAR + index of the variable
  INTEGER, PARAMETER :: NFUN = 14
  CHARACTER(LEN=5), PARAMETER :: FUNS(NFUN) = [CHARACTER(LEN=5) :: 'abs', 'exp', 'log10', 'log', 'sqrt', &
       'sinh', 'cosh', 'tanh', 'sin', 'cos', 'tan', 'asin', 'acos', 'atan']
  CHARACTER(LEN=5), PARAMETER :: BINOPS = '+-*/^'

  TYPE EXPRESSION
     INTEGER :: NCODE = 0, NIMM = 0
     INTEGER, ALLOCATABLE :: CODE(:)
     DOUBLE PRECISION, ALLOCATABLE :: IMM(:)
     LOGICAL :: VALID = .FALSE.
     CHARACTER(LEN=:), ALLOCATABLE :: TEXT
  END TYPE EXPRESSION

CONTAINS

  SUBROUTINE EXPR_COMPILE(EX, STR, VARS)
    TYPE(EXPRESSION), INTENT(OUT) :: EX
    CHARACTER(LEN=*), INTENT(IN) :: STR
    CHARACTER(LEN=*), INTENT(IN) :: VARS(:)
    CHARACTER(LEN=:), ALLOCATABLE :: S
    LOGICAL, ALLOCATABLE :: BIN(:)
    INTEGER :: I, N
    LOGICAL :: OK

    ! strip blanks, '**' -> '^'
    S = ''
    I = 1
    N = LEN_TRIM(STR)
    DO WHILE (I <= N)
       IF (STR(I:I) == ' ' .OR. STR(I:I) == ACHAR(9)) THEN
          I = I + 1
       ELSEIF (I < N .AND. STR(I:MIN(I + 1, N)) == '**') THEN
          S = S // '^'
          I = I + 2
       ELSE
          S = S // STR(I:I)
          I = I + 1
       ENDIF
    ENDDO
    EX%TEXT = S
    N = LEN(S)
    ALLOCATE(EX%CODE(MAX(2 * N + 4, 8)), EX%IMM(MAX(N, 4)))
    IF (N == 0) RETURN
    ALLOCATE(BIN(N))
    DO I = 1, N
       BIN(I) = IS_BINARY(S, I)
    ENDDO
    OK = .TRUE.
    CALL EMIT(1, N)
    EX%VALID = OK

  CONTAINS

    SUBROUTINE PUSH(C)
      INTEGER, INTENT(IN) :: C
      EX%NCODE = EX%NCODE + 1
      EX%CODE(EX%NCODE) = C
    END SUBROUTINE PUSH

    LOGICAL FUNCTION ENCLOSED(B, E)
      ! S(B:E) is '(' ... ')' with the two parentheses matching each other
      INTEGER, INTENT(IN) :: B, E
      INTEGER :: J, DEPTH
      ENCLOSED = .FALSE.
      IF (E <= B) RETURN
      IF (S(B:B) /= '(' .OR. S(E:E) /= ')') RETURN
      DEPTH = 0
      DO J = B + 1, E - 1
         IF (S(J:J) == '(') DEPTH = DEPTH + 1
         IF (S(J:J) == ')') DEPTH = DEPTH - 1
         IF (DEPTH < 0) RETURN
      ENDDO
      ENCLOSED = DEPTH == 0
    END FUNCTION ENCLOSED

    INTEGER FUNCTION FUN_AT(B, E)
      ! index of the function whose name starts S(B:E) (first match in list order)
      INTEGER, INTENT(IN) :: B, E
      INTEGER :: K, L
      FUN_AT = 0
      DO K = 1, NFUN
         L = LEN_TRIM(FUNS(K))
         IF (B + L - 1 > E) CYCLE
         IF (LOWER(S(B:B + L - 1)) == TRIM(FUNS(K))) THEN
            FUN_AT = K
            RETURN
         ENDIF
      ENDDO
    END FUNCTION FUN_AT

    RECURSIVE SUBROUTINE EMIT(B, E)
      INTEGER, INTENT(IN) :: B, E
      INTEGER :: J, K, DEPTH, CLS, B2, P
      IF (.NOT. OK) RETURN
      IF (B > E) THEN
         OK = .FALSE.
         RETURN
      ENDIF
      IF (S(B:B) == '+') THEN                      ! unary plus
         CALL EMIT(B + 1, E)
         RETURN
      ENDIF
      IF (ENCLOSED(B, E)) THEN
         CALL EMIT(B + 1, E - 1)
         RETURN
      ENDIF
      IF (IS_LETTER(S(B:B))) THEN                  ! fcn( ... )
         K = FUN_AT(B, E)
         IF (K > 0) THEN
            P = INDEX(S(B:E), '(')
            IF (P > 0) THEN
               B2 = B + P - 1
               IF (ENCLOSED(B2, E)) THEN
                  CALL EMIT(B2 + 1, E - 1)
                  CALL PUSH(OP_FUN + K)
                  RETURN
               ENDIF
            ENDIF
         ENDIF
      ELSEIF (S(B:B) == '-' .AND. B < E) THEN
         IF (ENCLOSED(B + 1, E)) THEN               ! -( ... )
            CALL EMIT(B + 2, E - 1)
            CALL PUSH(OP_NEG)
            RETURN
         ELSEIF (IS_LETTER(S(B + 1:B + 1))) THEN    ! -fcn( ... )
            K = FUN_AT(B + 1, E)
            IF (K > 0) THEN
               P = INDEX(S(B + 1:E), '(')
               IF (P > 0) THEN
                  B2 = B + P
                  IF (ENCLOSED(B2, E)) THEN
                     CALL EMIT(B2 + 1, E - 1)
                     CALL PUSH(OP_FUN + K)
                     CALL PUSH(OP_NEG)
                     RETURN
                  ENDIF
               ENDIF
            ENDIF
         ENDIF
      ENDIF
      ! rightmost top-level binary operator of the lowest class present
      DO CLS = 1, 5
         DEPTH = 0
         DO J = E, B, -1
            IF (S(J:J) == ')') DEPTH = DEPTH + 1
            IF (S(J:J) == '(') DEPTH = DEPTH - 1
            IF (DEPTH /= 0) CYCLE
            IF (S(J:J) /= BINOPS(CLS:CLS)) CYCLE
            IF (.NOT. BIN(J)) CYCLE
            IF (CLS >= 3 .AND. S(B:B) == '-') THEN   ! -a*b : negate the product
               CALL EMIT(B + 1, E)
               CALL PUSH(OP_NEG)
            ELSE
               CALL EMIT(B, J - 1)
               CALL EMIT(J + 1, E)
               CALL PUSH(OP_ADD + CLS - 1)
            ENDIF
            RETURN
         ENDDO
      ENDDO
      ! a single item: number or variable, possibly with a leading minus
      B2 = B
      IF (S(B:B) == '-') B2 = B + 1
      IF (B2 > E) THEN
         OK = .FALSE.
         RETURN
      ENDIF
      IF (SCAN(S(B2:B2), '0123456789.') > 0) THEN
         EX%NIMM = EX%NIMM + 1
         EX%IMM(EX%NIMM) = READ_NUMBER(S(B2:E), OK)
         CALL PUSH(OP_IMM)
      ELSE
         K = 0
         DO J = 1, SIZE(VARS)
            IF (S(B2:E) == TRIM(VARS(J))) THEN
               K = J
               EXIT
            ENDIF
         ENDDO
         IF (K == 0) THEN
            OK = .FALSE.
            RETURN
         ENDIF
         CALL PUSH(OP_VAR + K)
      ENDIF
      IF (B2 > B) CALL PUSH(OP_NEG)
    END SUBROUTINE EMIT

  END SUBROUTINE EXPR_COMPILE

  ! Is the operator character at S(J:J) a binary operator?  '+'/'-' are unary
  ! at the start or after another operator / '(' , and belong to a number when
  ! they sign the exponent of a real literal (1.5d-3).
  LOGICAL FUNCTION IS_BINARY(S, J)
    CHARACTER(LEN=*), INTENT(IN) :: S
    INTEGER, INTENT(IN) :: J
    INTEGER :: K
    LOGICAL :: DIGITS, POINT
    IS_BINARY = SCAN(S(J:J), '+-*/^') > 0
    IF (.NOT. IS_BINARY) RETURN
    IF (S(J:J) /= '+' .AND. S(J:J) /= '-') RETURN
    IF (J == 1) THEN
       IS_BINARY = .FALSE.
       RETURN
    ENDIF
    IF (SCAN(S(J - 1:J - 1), '+-*/^(') > 0) THEN
       IS_BINARY = .FALSE.
       RETURN
    ENDIF
    IF (J < LEN(S) .AND. J > 2) THEN
       IF (SCAN(S(J + 1:J + 1), '0123456789') > 0 .AND. SCAN(S(J - 1:J - 1), 'eEdD') > 0) THEN
          ! walk left over a mantissa: digits with at most one '.'
          DIGITS = .FALSE.
          POINT = .FALSE.
          K = J - 1
          DO WHILE (K > 1)
             K = K - 1
             IF (SCAN(S(K:K), '0123456789') > 0) THEN
                DIGITS = .TRUE.
             ELSEIF (S(K:K) == '.') THEN
                IF (POINT) EXIT
                POINT = .TRUE.
             ELSE
                EXIT
             ENDIF
          ENDDO
          IF (DIGITS .AND. (K == 1 .OR. SCAN(S(K:K), '+-*/^(') > 0)) IS_BINARY = .FALSE.
       ENDIF
    ENDIF
  END FUNCTION IS_BINARY

  DOUBLE PRECISION FUNCTION READ_NUMBER(T, OK)
    CHARACTER(LEN=*), INTENT(IN) :: T
    LOGICAL, INTENT(INOUT) :: OK
    INTEGER :: IOS
    READ(T, *, IOSTAT=IOS) READ_NUMBER
    IF (IOS /= 0) THEN
       READ_NUMBER = 0.0D0
       OK = .FALSE.
    ENDIF
  END FUNCTION READ_NUMBER

  LOGICAL FUNCTION IS_LETTER(C)
    CHARACTER(LEN=1), INTENT(IN) :: C
    IS_LETTER = (C >= 'a' .AND. C <= 'z') .OR. (C >= 'A' .AND. C <= 'Z')
  END FUNCTION IS_LETTER

  FUNCTION LOWER(T) RESULT(R)
    CHARACTER(LEN=*), INTENT(IN) :: T
    CHARACTER(LEN=LEN(T)) :: R
    INTEGER :: I
    R = T
    DO I = 1, LEN(T)
       IF (T(I:I) >= 'A' .AND. T(I:I) <= 'Z') R(I:I) = ACHAR(IACHAR(T(I:I)) + 32)
    ENDDO
  END FUNCTION LOWER

  ! Value of the expression for the variable values VAL (same order as VARS).
  DOUBLE PRECISION FUNCTION EXPR_EVAL(EX, VAL) RESULT(RES)
    TYPE(EXPRESSION), INTENT(IN) :: EX
    DOUBLE PRECISION, INTENT(IN) :: VAL(:)
    DOUBLE PRECISION :: ST(EX%NCODE + 1)
    INTEGER :: IP, SP, DP, C
    RES = 0.0D0
    IF (.NOT. EX%VALID) RETURN
    SP = 0
    DP = 0
    DO IP = 1, EX%NCODE
       C = EX%CODE(IP)
       SELECT CASE (C)
       CASE (OP_IMM)
          DP = DP + 1
          SP = SP + 1
          ST(SP) = EX%IMM(DP)
       CASE (OP_NEG)
          ST(SP) = -ST(SP)
       CASE (OP_ADD)
          ST(SP - 1) = ST(SP - 1) + ST(SP)
          SP = SP - 1
       CASE (OP_SUB)
          ST(SP - 1) = ST(SP - 1) - ST(SP)
          SP = SP - 1
       CASE (OP_MUL)
          ST(SP - 1) = ST(SP - 1) * ST(SP)
          SP = SP - 1
       CASE (OP_DIV)
          IF (ST(SP) == 0.0D0) RETURN
          ST(SP - 1) = ST(SP - 1) / ST(SP)
          SP = SP - 1
       CASE (OP_POW)
          ST(SP - 1) = ST(SP - 1)**ST(SP)
          SP = SP - 1
       CASE (OP_FUN + 1)
          ST(SP) = ABS(ST(SP))
       CASE (OP_FUN + 2)
          ST(SP) = EXP(ST(SP))
       CASE (OP_FUN + 3)
          IF (ST(SP) <= 0.0D0) RETURN
          ST(SP) = LOG10(ST(SP))
       CASE (OP_FUN + 4)
          IF (ST(SP) <= 0.0D0) RETURN
          ST(SP) = LOG(ST(SP))
       CASE (OP_FUN + 5)
          IF (ST(SP) < 0.0D0) RETURN
          ST(SP) = SQRT(ST(SP))
       CASE (OP_FUN + 6)
          ST(SP) = SINH(ST(SP))
       CASE (OP_FUN + 7)
          ST(SP) = COSH(ST(SP))
       CASE (OP_FUN + 8)
          ST(SP) = TANH(ST(SP))
       CASE (OP_FUN + 9)
          ST(SP) = SIN(ST(SP))
       CASE (OP_FUN + 10)
          ST(SP) = COS(ST(SP))
       CASE (OP_FUN + 11)
          ST(SP) = TAN(ST(SP))
       CASE (OP_FUN + 12)
          IF (ABS(ST(SP)) > 1.0D0) RETURN
          ST(SP) = ASIN(ST(SP))
       CASE (OP_FUN + 13)
          IF (ABS(ST(SP)) > 1.0D0) RETURN
          ST(SP) = ACOS(ST(SP))
       CASE (OP_FUN + 14)
          ST(SP) = ATAN(ST(SP))
       CASE DEFAULT
          SP = SP + 1
          ST(SP) = VAL(C - OP_VAR)
       END SELECT
    ENDDO
    IF (SP >= 1) RES = ST(1)
  END FUNCTION EXPR_EVAL

END MODULE KFSP_EXPR
