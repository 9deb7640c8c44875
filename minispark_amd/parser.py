"""SQL text -> DataFrame (reference: src/mini_spark/parser.py - the grammar at :14-69 and what its visitor builds
at :124-162; the reference parses with the third-party `parsimonious`, which is not part of this package's
requirements, so `.sql()` would not work next to the HIP engine without this module).

A hand-written backtracking recursive-descent parser for the same language:

    SELECT select_list FROM 'path' [AS t] { [LEFT|RIGHT|INNER|FULL] JOIN 'path' [AS t] ON cond }
           [WHERE cond] [GROUP BY col [HAVING cond]] ;

* select items: ``*``, ``COUNT()/SUM(e)/AVG(e)/MIN(e)/MAX(e) [AS name]``, ``expr [AS name]``;
* conditions: OR < AND < NOT < comparison | ( cond ) | BETWEEN | LIKE, comparators = != <= >= < >;
* expressions: + - over * / over atoms (number, column, string literal, ( expr ), COUNT()/SUM(e));
* like the reference: whitespace is required around keywords, the closing ``;`` is mandatory, every join kind
  is executed as an inner join (parser.py:131-133), numbers are integers (parser.py:349), NOT raises
  NotImplementedError (sql.py:44-45), GROUP BY takes one column (dataframe.py:64).

Alternatives are tried in the grammar's order and the first that fits wins (ordered choice), so texts the
reference accepts build the same task tree here (tests/test_parser.py compares both renderings).
"""

from __future__ import annotations

import operator
import re
from typing import Any, Callable

from .dataframe import DataFrame
from .sql import AggCol, Col, Lit
from .sql import Functions as F


class SqlSyntaxError(ValueError):
    """The text is not a query of the supported language (the reference raises parsimonious' ParseError)."""


class SemanticError(Exception):
    pass


class GroupByError(SemanticError):
    pass


class _NoMatch(Exception):
    pass


_WS = re.compile(r"\s+")
_TABLE = re.compile(r"[a-zA-Z0-9_\-\./ ]+")
_COLUMN = re.compile(r"[A-Za-z_][A-Za-z0-9_\.]*")
_IDENT = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")
_NUMBER = re.compile(r"-?[0-9]+(\.[0-9]+)?")
_STRING = re.compile(r"[^']*")
_COMPARATORS: list[tuple[str, Callable[[Any, Any], Any]]] = [
    ("=", operator.eq), ("!=", operator.ne), ("<=", operator.le), (">=", operator.ge), ("<", operator.lt),
    (">", operator.gt),
]
_JOIN_KINDS = [("JOIN",), ("LEFT", "JOIN"), ("RIGHT", "JOIN"), ("INNER", "JOIN"), ("FULL", "JOIN")]
_AGGREGATES = ("COUNT", "SUM", "AVG", "MIN", "MAX")


class _Parser:
    def __init__(self, text: str, engine: Any) -> None:
        self.text, self.pos, self.engine = text, 0, engine
        self.furthest = 0

    # ---- matching primitives --------------------------------------------------------------------------------
    def fail(self) -> None:
        self.furthest = max(self.furthest, self.pos)
        raise _NoMatch

    def lit(self, s: str) -> str:
        if not self.text.startswith(s, self.pos):
            self.fail()
        self.pos += len(s)
        return s

    def rx(self, pattern: re.Pattern) -> str:
        m = pattern.match(self.text, self.pos)
        if m is None or m.end() == m.start():
            self.fail()
        self.pos = m.end()
        return m.group(0)

    def ws(self) -> None:
        self.rx(_WS)

    def ows(self) -> None:
        m = _WS.match(self.text, self.pos)
        if m:
            self.pos = m.end()

    def attempt(self, rule: Callable[[], Any]) -> tuple[bool, Any]:
        start = self.pos
        try:
            return True, rule()
        except _NoMatch:
            self.pos = start
            return False, None

    def first_of(self, *rules: Callable[[], Any]) -> Any:
        for rule in rules:
            ok, value = self.attempt(rule)
            if ok:
                return value
        self.fail()
        return None

    def repeat(self, rule: Callable[[], Any]) -> list[Any]:
        out = []
        while True:
            ok, value = self.attempt(rule)
            if not ok:
                return out
            out.append(value)

    # ---- query ------------------------------------------------------------------------------------------------
    def query(self) -> DataFrame:
        self.ows()
        self.lit("SELECT")
        self.ws()
        select_list = self.select_list()
        self.ws()
        self.lit("FROM")
        self.ws()
        df = self.table_reference()
        joins = self.repeat(lambda: (self.ws(), self.join_clause())[1])
        has_where, where = self.attempt(lambda: (self.ws(), self.where_clause())[1])
        has_group, group = self.attempt(lambda: (self.ws(), self.group_by_clause())[1])
        self.ows()
        self.lit(";")
        self.ows()
        if self.pos != len(self.text):
            self.fail()

        for other, cond in joins:
            df = df.join(other, on=cond, how="inner")
        if has_where:
            df = df.filter(where)
        if has_group:
            group_cols, having = group
            key_names = {c.name for c in group_cols}
            agg_cols = [c for c in select_list if type(c) is AggCol]
            stray = [c for c in select_list if type(c) is not AggCol and c.name not in key_names]
            if stray:
                raise GroupByError(
                    "All selected columns must be aggregate functions or part of the key when using GROUP BY:\n"
                    f"{stray}"
                )
            if having is not None:
                extra = [c for c in having.all_nested_columns if type(c) is AggCol]
                for c in extra:
                    c.name = f"_having_{c.name}"
                agg_cols.extend(extra)
            df = df.group_by(*group_cols).agg(*agg_cols)
            if having is not None:
                df = df.filter(having.normalize_agg_columns())
            return df.select(*[Col(c.name) for c in select_list])
        return df.select(*select_list)

    def select_list(self) -> list[Col]:
        items = [self.select_item()]
        items += self.repeat(lambda: (self.ows(), self.lit(","), self.ows(), self.select_item())[3])
        return items

    def select_item(self) -> Col:
        return self.first_of(self.star, self.aggregate_call, self.expr_aliased)

    def star(self) -> Col:
        self.lit("*")
        return Col("*")

    def aggregate_call(self) -> AggCol:
        name = self.first_of(*[(lambda n=n: self.lit(n)) for n in _AGGREGATES])
        self.lit("(")
        has_arg, arg = self.attempt(self.expr)
        self.lit(")")
        has_alias, alias = self.attempt(self.alias)
        if name == "COUNT":
            if has_arg:
                raise AssertionError("COUNT takes no argument")
            agg = F.count()
        else:
            if not has_arg:
                raise AssertionError(f"{name} takes one argument")
            agg = {"SUM": F.sum, "AVG": F.avg, "MIN": F.min, "MAX": F.max}[name](_as_col(arg))
        return agg.alias(alias) if has_alias else agg

    def expr_aliased(self) -> Col:
        col = self.expr()
        has_alias, alias = self.attempt(self.alias)
        return _as_col(col).alias(alias) if has_alias else col

    def alias(self) -> str:
        self.ws()
        self.lit("AS")
        self.ws()
        return self.rx(_IDENT)

    def table_reference(self) -> DataFrame:
        self.lit("'")
        path = self.rx(_TABLE)
        self.lit("'")
        df = DataFrame(self.engine).table(path)
        has_alias, alias = self.attempt(self.alias)
        return df.alias(alias) if has_alias else df

    def join_clause(self) -> tuple[DataFrame, Col]:
        def kind(words: tuple[str, ...]) -> Callable[[], None]:
            def rule() -> None:
                self.lit(words[0])
                for w in words[1:]:
                    self.ws()
                    self.lit(w)
            return rule

        self.first_of(*[kind(words) for words in _JOIN_KINDS])
        self.ws()
        table = self.table_reference()
        self.ws()
        self.lit("ON")
        self.ws()
        return table, self.condition()

    def where_clause(self) -> Col:
        self.lit("WHERE")
        self.ws()
        return self.condition()

    def group_by_clause(self) -> tuple[list[Col], Col | None]:
        self.lit("GROUP")
        self.ws()
        self.lit("BY")
        self.ws()
        cols = [self.column_name()]
        cols += self.repeat(lambda: (self.ows(), self.lit(","), self.ows(), self.column_name())[3])

        def having() -> Col:
            self.ws()
            self.lit("HAVING")
            self.ws()
            return self.condition()

        has_having, cond = self.attempt(having)
        return cols, cond if has_having else None

    # ---- conditions -------------------------------------------------------------------------------------------
    def condition(self) -> Col:
        return self.or_expr()

    def or_expr(self) -> Col:
        left = self.and_expr()
        for right in self.repeat(lambda: (self.ws(), self.lit("OR"), self.ws(), self.and_expr())[3]):
            left = left | right
        return left

    def and_expr(self) -> Col:
        left = self.not_expr()
        for right in self.repeat(lambda: (self.ws(), self.lit("AND"), self.ws(), self.not_expr())[3]):
            left = left & right
        return left

    def not_expr(self) -> Col:
        negated, _ = self.attempt(lambda: (self.lit("NOT"), self.ws()))
        pred = self.predicate()
        return ~pred if negated else pred

    def predicate(self) -> Any:
        return self.first_of(self.comparison, self.parenthised_condition, self.string_literal, self.between, self.like)

    def comparison(self) -> Col:
        left = self.expr()
        self.ows()
        fn = self.first_of(*[(lambda s=s, f=f: (self.lit(s), f)[1]) for s, f in _COMPARATORS])
        self.ows()
        right = self.expr()
        return fn(left, right)

    def parenthised_condition(self) -> Col:
        self.lit("(")
        self.ows()
        cond = self.condition()
        self.ows()
        self.lit(")")
        return cond

    def between(self) -> Col:
        col = self.column_name()
        self.ws()
        self.lit("BETWEEN")
        self.ws()
        start = self.first_of(self.string_literal, self.column_name)
        self.ws()
        self.lit("AND")
        self.ws()
        end = self.first_of(self.string_literal, self.column_name)
        return col.between(start, end)

    def like(self) -> Col:
        col = self.expr()
        self.ws()
        self.lit("LIKE")
        self.ws()
        return _as_col(col).like(self.string_literal())

    # ---- arithmetic ---------------------------------------------------------------------------------------------
    def expr(self) -> Any:
        return self.add_expr()

    def _binary_chain(self, operand: Callable[[], Any], ops: dict[str, Callable[[Any, Any], Any]]) -> Any:
        left = operand()

        def tail() -> tuple[Callable[[Any, Any], Any], Any]:
            self.ows()
            fn = self.first_of(*[(lambda s=s, f=f: (self.lit(s), f)[1]) for s, f in ops.items()])
            self.ows()
            return fn, operand()

        for fn, right in self.repeat(tail):
            left = fn(left, right)
        return left

    def add_expr(self) -> Any:
        return self._binary_chain(self.mul_expr, {"+": operator.add, "-": operator.sub})

    def mul_expr(self) -> Any:
        return self._binary_chain(self.atom, {"*": operator.mul, "/": operator.truediv})

    def atom(self) -> Any:
        return self.first_of(self.function_call, self.number, self.column_name, self.parenthised_expr,
                             self.string_literal)

    def function_call(self) -> Col:
        name = self.rx(_IDENT)
        self.ows()
        self.lit("(")
        self.ows()

        def arguments() -> list[Any]:
            args = [self.expr()]
            args += self.repeat(lambda: (self.ows(), self.lit(","), self.ows(), self.expr())[3])
            return args

        has_args, args = self.attempt(arguments)
        self.ows()
        self.lit(")")
        args = args if has_args else []
        if name == "COUNT":
            if args:
                raise AssertionError("COUNT takes no argument")
            return F.count()
        if name == "SUM":
            if len(args) != 1:
                raise AssertionError("SUM takes one argument")
            return F.sum(_as_col(args[0]))
        raise SemanticError(f"Unsupported function: {name}")

    def parenthised_expr(self) -> Any:
        self.lit("(")
        self.ows()
        inner = self.expr()
        self.ows()
        self.lit(")")
        return inner

    # ---- terminals ----------------------------------------------------------------------------------------------
    def column_name(self) -> Col:
        return Col(self.rx(_COLUMN))

    def number(self) -> Lit:
        return Lit(int(self.rx(_NUMBER)))  # "1.5" -> ValueError, as in the reference (parser.py:349)

    def string_literal(self) -> str:
        self.lit("'")
        m = _STRING.match(self.text, self.pos)
        self.pos = m.end()
        self.lit("'")
        return m.group(0)


def _as_col(value: Any) -> Col:
    """A bare string literal used where a column expression is needed (e.g. SUM('x')) becomes a literal."""
    return value if isinstance(value, Col) else Lit(value)


def parse_sql(sql: str, engine: Any = None) -> DataFrame:
    """-> DataFrame bound to ``engine`` (None: the HIP engine is created on first use)."""
    parser = _Parser(sql, engine)
    try:
        return parser.query()
    except _NoMatch:
        at = parser.furthest
        raise SqlSyntaxError(f"not a supported query: cannot continue at offset {at}: {sql[at: at + 30]!r}") from None
