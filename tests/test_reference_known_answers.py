"""CPU: the reference's 9x9 gtest known answers (tests/known_answers.py) against the C restatement."""
import pytest

import known_answers as ka
from adapters import PortState
from pyoracle import Port


@pytest.fixture(scope="module")
def port9(built):
    return Port(9)


@pytest.mark.parametrize("case", ka.ALL_CASES, ids=lambda f: f.__name__)
def test_known_answer(port9, case):
    case(lambda: PortState(port9))
